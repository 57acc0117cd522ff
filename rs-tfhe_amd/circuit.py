"""Levelised gate circuits with device-resident ciphertexts.

The reference's real workloads are gate DAGs evaluated one gate at a time on the CPU
(examples/add_two_numbers.rs:11-50: full_adder / add).  Here a circuit is built once as a DAG
over wires, levelised, and every level -- all of its gates, whatever their types, times the
whole batch of independent inputs -- is ONE `tfhe_hip_batch_gates_mixed_dev` launch.  Wires
stay in HBM between levels; only the operand gather (an index_select) sits between launches.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import engine as E


@dataclass
class _Gate:
    op: int
    a: int
    b: int
    out: int
    level: int


@dataclass
class Circuit:
    n_inputs: int
    gates: list = field(default_factory=list)
    _level: dict = field(default_factory=dict)
    _n_wires: int = 0
    _plans: dict = field(default_factory=dict)  # (device, n_gates) -> per-level index / gate-code tensors

    def __post_init__(self):
        self._n_wires = self.n_inputs
        for w in range(self.n_inputs):
            self._level[w] = 0

    # -- construction ----------------------------------------------------------------
    def gate(self, op: int, a: int, b: int) -> int:
        """A bootstrapped two-input gate (one of tfhe_hip_gate); returns its output wire."""
        out = self._n_wires
        self._n_wires += 1
        lvl = 1 + max(self._level[a], self._level[b])
        self._level[out] = lvl
        self.gates.append(_Gate(op, a, b, out, lvl))
        return out

    def nand(self, a, b): return self.gate(E.NAND, a, b)
    def and_(self, a, b): return self.gate(E.AND, a, b)
    def or_(self, a, b): return self.gate(E.OR, a, b)
    def xor(self, a, b): return self.gate(E.XOR, a, b)
    def xnor(self, a, b): return self.gate(E.XNOR, a, b)
    def nor(self, a, b): return self.gate(E.NOR, a, b)
    def and_ny(self, a, b): return self.gate(E.ANDNY, a, b)
    def and_yn(self, a, b): return self.gate(E.ANDYN, a, b)
    def or_ny(self, a, b): return self.gate(E.ORNY, a, b)
    def or_yn(self, a, b): return self.gate(E.ORYN, a, b)

    def mux_naive(self, a, b, c):
        """Gates::mux_naive (src/gates.rs:189-199): or(and(a, b), and(not(a), c))."""
        return self.or_(self.and_(a, b), self.and_ny(a, c))

    def full_adder(self, a, b, c):
        """examples/add_two_numbers.rs:11-29 -> (sum, carry)."""
        a_xor_b = self.xor(a, b)
        a_and_b = self.and_(a, b)
        a_xor_b_and_c = self.and_(a_xor_b, c)
        s = self.xor(a_xor_b, c)
        carry = self.or_(a_and_b, a_xor_b_and_c)
        return s, carry

    def add(self, a_bits, b_bits, cin):
        """examples/add_two_numbers.rs:31-50 -> (sum bits, carry out)."""
        assert len(a_bits) == len(b_bits), "Cannot add two numbers with different number of bits!"
        result, carry = [], cin
        for x, y in zip(a_bits, b_bits):
            s, carry = self.full_adder(x, y, carry)
            result.append(s)
        return result, carry

    # -- schedule ----------------------------------------------------------------------
    @property
    def n_wires(self) -> int:
        return self._n_wires

    def levels(self):
        """Gates grouped by level (all operands of level L come from levels < L)."""
        depth = max((g.level for g in self.gates), default=0)
        out = [[] for _ in range(depth)]
        for g in self.gates:
            out[g.level - 1].append(g)
        return out

    # -- execution -----------------------------------------------------------------------
    def run_dev(self, eng, inputs, stream=None):
        """inputs: int32 CUDA tensor [n_inputs][B][n+1]; returns the wire store [n_wires][B][n+1]
        (int32 CUDA tensor, rows of non-existent wires undefined).  One launch per level.
        `eng`: an Engine, or a Pool -- then the wires live on the pool's home member (`pool.home`) and every level is
        one `tfhe_hip_pool_batch_gates_mixed_dev` call: its gates x batch are cut over the members, the shards travel
        by grouped RCCL send / receive and the level's results are back on the home GPU, in order, for the next gather."""
        import torch

        n_in, B, w = inputs.shape
        assert n_in == self.n_inputs
        # the gathers, the temporaries' allocations and the engine's kernels must share ONE stream: make `stream`
        # torch's current stream for the duration (a kernel on another stream would race the index_select that
        # feeds it, and the caching allocator could hand a temporary to someone else while it is still in use)
        with _on_stream(stream):
            wires = torch.empty((self.n_wires, B, w), dtype=torch.int32, device=inputs.device)
            wires[:n_in] = inputs
            for ia, ib, io, codes in self._plan(inputs.device):
                a = wires.index_select(0, ia).reshape(-1, w)
                b = wires.index_select(0, ib).reshape(-1, w)
                gc = codes.repeat_interleave(B).contiguous()
                out = torch.empty_like(a)
                eng.batch_gates_mixed_dev(gc, a, b, out)  # torch's current stream = `stream`
                wires.index_copy_(0, io, out.reshape(len(io), B, w))
        return wires

    def _plan(self, device):
        """Per-level operand / result wire indices and gate codes as device tensors, built once per
        (device, circuit size): repeated runs of the same circuit upload nothing but their inputs."""
        import torch

        key = (str(device), len(self.gates))
        plan = self._plans.get(key)
        if plan is None:
            plan = []
            for lvl in self.levels():
                plan.append((torch.tensor([g.a for g in lvl], device=device),
                             torch.tensor([g.b for g in lvl], device=device),
                             torch.tensor([g.out for g in lvl], device=device),
                             torch.tensor([g.op for g in lvl], dtype=torch.uint8, device=device)))
            self._plans[key] = plan
        return plan

    def run(self, eng, inputs) -> np.ndarray:
        """Host convenience: inputs uint32 [n_inputs][B][n+1] -> all wires uint32 [n_wires][B][n+1].  `eng`: Engine or Pool."""
        import torch

        dev = torch.device("cuda", eng.device)
        t = torch.from_numpy(np.ascontiguousarray(inputs, dtype=np.uint32).view(np.int32)).to(dev)
        with torch.cuda.device(dev):
            wires = self.run_dev(eng, t)
            if isinstance(eng, E.Pool):
                eng.synchronize()
            torch.cuda.synchronize()
        return wires.cpu().numpy().view(np.uint32)

    def run_reference(self, gate_fn, inputs) -> np.ndarray:
        """Evaluate gate by gate with `gate_fn(op, a[B][n+1], b[B][n+1]) -> [B][n+1]`: the order
        the reference's example executes it in (the tests plug their CPU checker in here)."""
        inputs = np.ascontiguousarray(inputs, dtype=np.uint32)
        wires = np.zeros((self.n_wires,) + inputs.shape[1:], np.uint32)
        wires[: self.n_inputs] = inputs
        for g in self.gates:
            wires[g.out] = gate_fn(g.op, wires[g.a], wires[g.b])
        return wires


def _on_stream(stream):
    """torch.cuda.stream(stream) when a stream is given, a no-op context otherwise."""
    import contextlib

    import torch

    return torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()


# ---- LUT arithmetic: the nibble adder of examples/lut_add_two_numbers.rs, batched ---------------
def lut_add_u8_dev(eng, a_low, a_high, b_low, b_high, stream=None):
    """8-bit addition with three programmable bootstraps per byte pair instead of eight gate
    bootstraps per bit pair (examples/lut_add_two_numbers.rs:82-158), for a whole batch, on the device.

    Inputs: int32 CUDA tensors [count][n+1], encryptions of the low / high nibbles of a and b under
    message modulus 32 (`encrypt_lwe_message(nibble, 32, ...)`, :99-122).  Returns
    (sum_low, sum_high, carry) as device tensors, each decrypting (modulus 32) to the nibble / bit.

    Every TLWE addition the example performs with `&x + &y` is folded into the prologue of the
    bootstrap that consumes it (tfhe_hip_batch_lincomb_bootstrap_dev), or is one streaming kernel
    (tfhe_hip_batch_tlwe_lincomb_dev) for the three-operand high sum; nothing leaves HBM."""
    import torch

    from .lut import Generator

    gen = Generator(32)  # message modulus 32 covers every possible nibble sum 0..30 (:86-87)
    dev = a_low.device
    with _on_stream(stream):  # uploads, allocations and kernels on one stream (see Circuit.run_dev)
        lut_mod16 = torch.from_numpy(gen.generate_lookup_table(lambda x: x % 16).poly.view(np.int32)).to(dev)
        lut_carry = torch.from_numpy(gen.generate_lookup_table(lambda x: 1 if x >= 16 else 0).poly.view(np.int32)).to(dev)
        sum_low, carry = torch.empty_like(a_low), torch.empty_like(a_low)
        high, sum_high = torch.empty_like(a_low), torch.empty_like(a_low)
        # bootstraps 1 and 2: low sum mod 16 and its carry, both from a_low + b_low (:124-150)
        eng.batch_lincomb_bootstrap_dev(1, a_low, 1, b_low, 0, sum_low, testvec=lut_mod16)
        eng.batch_lincomb_bootstrap_dev(1, a_low, 1, b_low, 0, carry, testvec=lut_carry)
        # a_high + b_high (:152-153), then bootstrap 3 on (a_high + b_high) + carry (:155-158)
        eng.batch_tlwe_lincomb_dev(1, a_high, 1, b_high, 0, high)
        eng.batch_lincomb_bootstrap_dev(1, high, 1, carry, 0, sum_high, testvec=lut_mod16)
    return sum_low, sum_high, carry


def mux_and_gates_dev(eng, a, b, c, codes, xa, xb, stream=None):
    """One circuit level holding `M` Gates::mux (the reference's formula, src/gates.rs:157-183) beside `X` two-input
    gates (`codes`: uint8 device tensor [X]) -- BASELINE configs[4] is M hom_mux + X hom_xor -- in TWO blind-rotation
    launches and ONE key switch whatever M and X:
      launch 1  [and(a, b) | and(not(a), c)] for all M, bootstrap_without_key_switch      (gates.rs:165-177)
      launch 2  [or(u1, u2) for all M | the X other gates], full bootstrap                   (gates.rs:179-182)
    a, b, c: int32 CUDA tensors [M][n+1]; xa, xb: [X][n+1].  Returns (mux_out [M][n+1], gate_out [X][n+1]).
    `eng`: an Engine, or a Pool (tensors on its home member's GPU): each of the two launches is then a pool call cut
    over the members -- configs[4]'s level through ONE handle."""
    import torch

    M, X = a.shape[0], xa.shape[0]
    with _on_stream(stream):
        g1 = torch.empty(2 * M, dtype=torch.uint8, device=a.device)
        g1[:M] = E.AND
        g1[M:] = E.ANDNY
        u = torch.empty((2 * M, a.shape[1]), dtype=torch.int32, device=a.device)
        eng.batch_gates_mixed_dev(g1, torch.cat([a, a]), torch.cat([b, c]), u, keyswitch=False)
        g2 = torch.cat([torch.full((M,), E.OR, dtype=torch.uint8, device=a.device), codes])
        out = torch.empty((M + X, a.shape[1]), dtype=torch.int32, device=a.device)
        eng.batch_gates_mixed_dev(g2, torch.cat([u[:M], xa]), torch.cat([u[M:], xb]), out)
    return out[:M], out[M:]


def lut_add_u8(eng, a_low, a_high, b_low, b_high):
    """Host-array form of lut_add_u8_dev: numpy [count][n+1] in, (sum_low, sum_high, carry) out."""
    from .lut import Generator

    gen = Generator(32)
    lut_mod16 = gen.generate_lookup_table(lambda x: x % 16).poly
    lut_carry = gen.generate_lookup_table(lambda x: 1 if x >= 16 else 0).poly
    sum_low = eng.batch_lincomb_bootstrap(1, a_low, 1, b_low, testvec=lut_mod16)
    carry = eng.batch_lincomb_bootstrap(1, a_low, 1, b_low, testvec=lut_carry)
    high = eng.batch_tlwe_lincomb(1, a_high, 1, b_high)
    sum_high = eng.batch_lincomb_bootstrap(1, high, 1, carry, testvec=lut_mod16)
    return sum_low, sum_high, carry
