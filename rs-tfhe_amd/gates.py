"""Homomorphic gates behind the reference's `Gates` API (src/gates.rs).

Single gates are the batch path with count == 1; batch_* mirror
gates::batch_{nand,and,or,xor,nor,xnor} (src/gates.rs:352-547) and take the
two operand batches as [count][n+1] arrays (the reference's slice of pairs).
The linear prep of every gate is fused into the blind-rotate kernel.
"""
from __future__ import annotations

import numpy as np

from . import engine as E
from .bootstrap import Bootstrap, HipBootstrap, default_bootstrap, keyed_engine
from .params import f64_to_torus


def _gate(gate: int, a, b, cloud_key, device: int = 0):
    a = np.asarray(a, dtype=np.uint32)
    with keyed_engine(cloud_key, device) as eng:  # key choice + launch: one critical section
        out = eng.batch_gate(gate, a, b)
    return out[0] if a.ndim == 1 else out


class Gates:
    """src/gates.rs:30-219"""

    def __init__(self, bootstrap: Bootstrap | None = None):
        self.bootstrap = bootstrap if bootstrap is not None else default_bootstrap()

    @classmethod
    def with_bootstrap(cls, bootstrap: Bootstrap) -> "Gates":  # gates.rs:43-45
        return cls(bootstrap)

    def bootstrap_strategy(self) -> str:  # gates.rs:48-50
        return self.bootstrap.name()

    def _fused(self) -> bool:
        # the fused prep+bootstrap kernel is the plain HipBootstrap strategy
        return type(self.bootstrap) is HipBootstrap

    def _run(self, gate: int, ca: int, cb: int, const: float, a, b, cloud_key):
        if self._fused():
            return _gate(gate, a, b, cloud_key, self.bootstrap.device)
        # any other strategy: do the linear prep here, then its bootstrap()
        a = np.asarray(a, dtype=np.uint32)
        b = np.asarray(b, dtype=np.uint32)
        t = (np.uint32(ca & 0xFFFFFFFF) * a + np.uint32(cb & 0xFFFFFFFF) * b).astype(np.uint32)
        t[..., -1] += np.uint32(f64_to_torus(const))
        return self.bootstrap.bootstrap(t, cloud_key)

    def nand(self, a, b, cloud_key):  # gates.rs:54-58
        return self._run(E.NAND, -1, -1, 0.125, a, b, cloud_key)

    def or_(self, a, b, cloud_key):  # gates.rs:62-66
        return self._run(E.OR, 1, 1, 0.125, a, b, cloud_key)

    def and_(self, a, b, cloud_key):  # gates.rs:70-74
        return self._run(E.AND, 1, 1, -0.125, a, b, cloud_key)

    def xor(self, a, b, cloud_key):  # gates.rs:78-82
        return self._run(E.XOR, 1, 2, 0.25, a, b, cloud_key)

    def xnor(self, a, b, cloud_key):  # gates.rs:86-90
        return self._run(E.XNOR, 1, -2, -0.25, a, b, cloud_key)

    def nor(self, a, b, cloud_key):  # gates.rs:94-98
        return self._run(E.NOR, -1, -1, -0.125, a, b, cloud_key)

    def and_ny(self, a, b, cloud_key):  # gates.rs:102-111
        return self._run(E.ANDNY, -1, 1, -0.125, a, b, cloud_key)

    def and_yn(self, a, b, cloud_key):  # gates.rs:115-124
        return self._run(E.ANDYN, 1, -1, -0.125, a, b, cloud_key)

    def or_ny(self, a, b, cloud_key):  # gates.rs:128-137
        return self._run(E.ORNY, -1, 1, 0.125, a, b, cloud_key)

    def or_yn(self, a, b, cloud_key):  # gates.rs:141-150
        return self._run(E.ORYN, 1, -1, 0.125, a, b, cloud_key)

    def mux(self, a, b, c, cloud_key):  # gates.rs:157-183 (reference formula)
        a = np.asarray(a, dtype=np.uint32)
        with keyed_engine(cloud_key, getattr(self.bootstrap, "device", 0)) as eng:
            out = eng.batch_mux(a, b, c, naive=False)
        return out[0] if a.ndim == 1 else out

    def mux_naive(self, a, b, c, cloud_key):  # gates.rs:189-199
        a = np.asarray(a, dtype=np.uint32)
        with keyed_engine(cloud_key, getattr(self.bootstrap, "device", 0)) as eng:
            out = eng.batch_mux(a, b, c, naive=True)
        return out[0] if a.ndim == 1 else out

    def not_(self, a):  # gates.rs:202-204 (no bootstrap)
        return (0 - np.asarray(a, dtype=np.uint32).astype(np.int64)).astype(np.uint32)

    def copy(self, a):  # gates.rs:207-209
        return np.array(a, dtype=np.uint32, copy=True)

    def constant(self, value: bool, n: int):  # gates.rs:212-219 (1 - mu wraps: quirk Q6)
        mu = f64_to_torus(0.125)
        mu = mu if value else (1 - mu) & 0xFFFFFFFF
        res = np.zeros(n + 1, np.uint32)
        res[n] = mu
        return res


# convenience free functions (src/gates.rs:233-326)
def nand(a, b, cloud_key): return Gates().nand(a, b, cloud_key)
def or_(a, b, cloud_key): return Gates().or_(a, b, cloud_key)
def and_(a, b, cloud_key): return Gates().and_(a, b, cloud_key)
def xor(a, b, cloud_key): return Gates().xor(a, b, cloud_key)
def xnor(a, b, cloud_key): return Gates().xnor(a, b, cloud_key)
def nor(a, b, cloud_key): return Gates().nor(a, b, cloud_key)
def and_ny(a, b, cloud_key): return Gates().and_ny(a, b, cloud_key)
def and_yn(a, b, cloud_key): return Gates().and_yn(a, b, cloud_key)
def or_ny(a, b, cloud_key): return Gates().or_ny(a, b, cloud_key)
def or_yn(a, b, cloud_key): return Gates().or_yn(a, b, cloud_key)
def mux(a, b, c, cloud_key): return Gates().mux(a, b, c, cloud_key)
def mux_naive(a, b, c, cloud_key): return Gates().mux_naive(a, b, c, cloud_key)
def not_(a): return Gates().not_(a)  # gates.rs:314-316
def copy(a): return Gates().copy(a)  # gates.rs:319-321
def constant(value: bool, n: int): return Gates().constant(value, n)  # gates.rs:324-326 (n: the ciphertexts' dimension)


# batch free functions (src/gates.rs:352-547); inputs_a/inputs_b: [count][n+1]
def batch_nand(inputs_a, inputs_b, cloud_key, device: int = 0): return _gate(E.NAND, inputs_a, inputs_b, cloud_key, device)
def batch_and(inputs_a, inputs_b, cloud_key, device: int = 0): return _gate(E.AND, inputs_a, inputs_b, cloud_key, device)
def batch_or(inputs_a, inputs_b, cloud_key, device: int = 0): return _gate(E.OR, inputs_a, inputs_b, cloud_key, device)
def batch_xor(inputs_a, inputs_b, cloud_key, device: int = 0): return _gate(E.XOR, inputs_a, inputs_b, cloud_key, device)
def batch_nor(inputs_a, inputs_b, cloud_key, device: int = 0): return _gate(E.NOR, inputs_a, inputs_b, cloud_key, device)
def batch_xnor(inputs_a, inputs_b, cloud_key, device: int = 0): return _gate(E.XNOR, inputs_a, inputs_b, cloud_key, device)


def batch_blind_rotate(srcs, cloud_key, device: int = 0):
    """trgsw::batch_blind_rotate (src/trgsw.rs:289-294): [count][n+1] -> [count][2][N]."""
    with keyed_engine(cloud_key, device) as eng:
        return eng.batch_blind_rotate(srcs)
