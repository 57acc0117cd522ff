"""GPU tests of the device-resident pool calls (`tfhe_hip_pool_batch_*_dev`, include/tfhe_hip.h): a batch that lives
on ONE member's GPU is cut over the members, the shards travel (grouped ncclSend / ncclRecv over the pool's persistent
communicator, or hipMemcpyPeerAsync behind events), every member bootstraps its shard, the results come back in input
order -- the reference's order-preserving `par_map` (src/parallel/rayon_impl.rs:40-47, src/gates.rs:357-383) for a
caller whose ciphertexts are already in HBM.

The test box has one GPU, so the pools here repeat device 0: `[0, 0]` and `[0] * 8` run the peer-copy transport
through exactly the code an 8-GPU pool runs (staging buffers, events, shard arithmetic, home rotation), and a pool of
ONE member under TFHE_HIP_POOL_RCCL=2 sends its shard to itself through ncclSend / ncclRecv, which exercises the RCCL
symbols, the group structure and the stream handling.  Results are held to the single-context path word for word.
"""
import numpy as np
import pytest

from conftest import oracle_keys

pytestmark = pytest.mark.gpu
N = 1024


def _cloud_key(ck):
    from test_gpu_parity import _cloud_key as ck_of

    return ck_of(ck)


def _dev(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to("cuda:0")


def _host(t):
    return t.cpu().numpy().view(np.uint32)


@pytest.fixture
def eng128(O, keys128):
    import rs_tfhe_amd as R

    sk, ck = keys128
    pk = _cloud_key(ck)
    eng = R.bootstrap.engine_for(pk.params, 0)
    eng.ensure_key(pk)
    return eng


def _check_all_entry_points(O, pool, eng, sk, counts, seed, homes=(0,)):
    """Every *_dev pool call against the single-context *_dev call on the same device tensors."""
    import torch

    rng = np.random.default_rng(seed)
    n1 = eng.params.n + 1
    for count in counts:
        A, B, Cc = (rng.integers(0, 2, count).astype(bool) for _ in range(3))
        ca, cb, cc = sk.encrypt_bool(A, seed + count), sk.encrypt_bool(B, seed + 1 + count), sk.encrypt_bool(Cc, seed + 2 + count)
        ta, tb, tc = _dev(ca), _dev(cb), _dev(cc)
        codes = rng.integers(0, 11, count).astype(np.uint8)
        tcodes = torch.from_numpy(codes).to("cuda:0")
        tv1 = _dev(rng.integers(0, 2**32, (2, N), dtype=np.uint64).astype(np.uint32))
        tvn = _dev(rng.integers(0, 2**32, (count, 2, N), dtype=np.uint64).astype(np.uint32))
        for home in homes:
            def both(call):
                want, got = torch.empty_like(ta), torch.full_like(ta, 0x5A5A5A5A)
                call(eng, want, {})
                call(pool, got, {"home": home})
                pool.synchronize()
                torch.cuda.synchronize()
                return _host(want), _host(got)

            w, g = both(lambda e, o, kw: e.batch_gate_dev(O.GATE_NAND, ta, tb, o, **kw))
            assert np.array_equal(g, w), (count, home, "gate")
            assert np.array_equal(sk.decrypt_bool(g), ~(A & B))
            w, g = both(lambda e, o, kw: e.batch_gates_mixed_dev(tcodes, ta, tb, o, **kw))
            assert np.array_equal(g, w), (count, home, "gates_mixed")
            w, g = both(lambda e, o, kw: e.batch_gates_mixed_dev(tcodes, ta, tb, o, keyswitch=False, **kw))
            assert np.array_equal(g, w), (count, home, "gates_mixed_nks")
            w, g = both(lambda e, o, kw: e.batch_bootstrap_dev(ta, o, **kw))
            assert np.array_equal(g, w), (count, home, "bootstrap")
            w, g = both(lambda e, o, kw: e.batch_bootstrap_dev(ta, o, testvec=tv1, keyswitch=False, **kw))  # shared table: sent whole to every peer
            assert np.array_equal(g, w), (count, home, "bootstrap shared testvec")
            w, g = both(lambda e, o, kw: e.batch_bootstrap_dev(ta, o, testvec=tvn, per_ct=True, **kw))  # per-ciphertext tables follow their shard
            assert np.array_equal(g, w), (count, home, "bootstrap per-ct testvec")
            w, g = both(lambda e, o, kw: e.batch_tlwe_lincomb_dev(3, ta, -2, tb, 0x1000, o, **kw))
            assert np.array_equal(g, w), (count, home, "tlwe_lincomb")
            assert np.array_equal(g, (3 * ca - 2 * cb + np.r_[np.zeros(n1 - 1, np.uint32), np.uint32(0x1000)]).astype(np.uint32))
            w, g = both(lambda e, o, kw: e.batch_lincomb_bootstrap_dev(1, ta, 1, tb, 0xE0000000, o, testvec=tv1, **kw))
            assert np.array_equal(g, w), (count, home, "lincomb_bootstrap")
            w, g = both(lambda e, o, kw: e.batch_mux_dev(ta, tb, tc, o, naive=True, **kw))
            assert np.array_equal(g, w), (count, home, "mux_naive")
            assert np.array_equal(sk.decrypt_bool(g), np.where(A, B, Cc))
            w, g = both(lambda e, o, kw: e.batch_mux_dev(ta, tb, tc, o, naive=False, **kw))
            assert np.array_equal(g, w), (count, home, "mux")
            want = torch.empty((count, 2, N), dtype=torch.int32, device="cuda:0")
            got = torch.zeros_like(want)
            eng.batch_blind_rotate_dev(ta, want)
            pool.batch_blind_rotate_dev(ta, got, home=home)
            pool.synchronize()
            torch.cuda.synchronize()
            assert np.array_equal(_host(got), _host(want)), (count, home, "blind_rotate")


def test_pool_resident_two_members_every_entry_point(O, eng128, keys128):
    """devices = {0, 0}: peer-copy transport; ragged counts below and above the 256-per-member cut, both homes."""
    import rs_tfhe_amd as R

    sk, ck = keys128
    pool = R.Pool(eng128.params, [0, 0])
    pool.load_cloud_key(_cloud_key(ck))
    _check_all_entry_points(O, pool, eng128, sk, counts=(1, 257, 601), seed=9100, homes=(0, 1))
    assert pool.data_transport == "peer-copy"
    # a batch too small to cut never leaves home
    import torch

    ca = sk.encrypt_bool(np.ones(5, bool), 9191)
    ta, to = _dev(ca), torch.empty((5, 701), dtype=torch.int32, device="cuda:0")
    pool.batch_gate_dev(O.GATE_NAND, ta, ta, to, home=1)
    assert pool.data_transport == "none"
    # argument checks: no such member, tensors of the wrong shape
    with pytest.raises(ValueError):
        pool.batch_gate_dev(O.GATE_NAND, ta, ta, to, home=2)
    with pytest.raises(R._capi.TfheHipError, match="home"):
        pool._chk(pool._lib.tfhe_hip_pool_batch_gate_dev(pool._h, 7, 0, ta.data_ptr(), ta.data_ptr(), to.data_ptr(), 5, None))
    with pytest.raises(R._capi.TfheHipError, match="unknown gate"):
        pool._chk(pool._lib.tfhe_hip_pool_batch_gate_dev(pool._h, 0, 99, ta.data_ptr(), ta.data_ptr(), to.data_ptr(), 5, None))
    pool.close()


def test_pool_resident_eight_members(O, eng128, keys128):
    """The eight-member shape of BASELINE configs[2] on one GPU (devices = [0] * 8): 8 contexts, 7 remote shards, home
    in the middle of the pool (shard r runs on member (home + r) mod 8), against the single context."""
    import torch

    import rs_tfhe_amd as R

    sk, ck = keys128
    pool = R.Pool(eng128.params, [0] * 8)
    pool.load_cloud_key(_cloud_key(ck))
    assert len(pool) == 8 and pool.members_for(2049) == 8
    rng = np.random.default_rng(9200)
    count = 2049 + 300  # 8 shards of 293 / 294
    A, B = rng.integers(0, 2, count).astype(bool), rng.integers(0, 2, count).astype(bool)
    ca, cb = sk.encrypt_bool(A, 9201), sk.encrypt_bool(B, 9202)
    ta, tb = _dev(ca), _dev(cb)
    want, got = torch.empty_like(ta), torch.zeros_like(ta)
    eng128.batch_gate_dev(O.GATE_NAND, ta, tb, want)
    pool.set_profiling(True)
    pool.batch_gate_dev(O.GATE_NAND, ta, tb, got, home=3)
    pool.synchronize()
    torch.cuda.synchronize()
    assert np.array_equal(_host(got), _host(want))
    assert np.array_equal(sk.decrypt_bool(_host(got)), ~(A & B))
    idx = np.linspace(0, count - 1, 48).astype(np.int64)  # and the oracle itself on a spread sample
    assert np.array_equal(_host(got)[idx], O.batch_gate(ck, O.GATE_NAND, ca[idx], cb[idx]))
    tt = pool.transfer_times()
    lo, hi = pool.shard(count, 0)
    assert tt["calls"] == 1 and tt["scatter_bytes"] == 2 * (count - (hi - lo)) * 701 * 4 and tt["gather_bytes"] == (count - (hi - lo)) * 701 * 4
    assert tt["scatter_ms_sum"] > 0 and tt["gather_ms_sum"] > 0
    pool.set_profiling(False)
    pool.close()


def test_pool_resident_rccl_loopback(O, eng128, keys128, monkeypatch):
    """TFHE_HIP_POOL_RCCL=2, a pool of ONE member: the member's own shard takes the remote path through a self
    ncclSend / ncclRecv on the pool's persistent communicator -- the RCCL symbols, the grouped calls and the stream
    order of scatter -> compute -> gather run for real (among >= 2 GPUs they cannot on this box)."""
    import torch

    import rs_tfhe_amd as R

    sk, ck = keys128
    monkeypatch.setenv("TFHE_HIP_POOL_RCCL", "2")
    pool = R.Pool(eng128.params, [0])
    pool.load_cloud_key(_cloud_key(ck))
    assert pool.key_transport == "rccl"
    _check_all_entry_points(O, pool, eng128, sk, counts=(3, 300), seed=9300)
    assert pool.data_transport == "rccl"
    # two calls back to back on a caller's stream: the second reads the first's output (stream order through the gather)
    s = torch.cuda.Stream()
    ca = sk.encrypt_bool(np.array([1, 0, 1, 1], bool), 9391)
    ta = _dev(ca)
    mid, out = torch.empty_like(ta), torch.empty_like(ta)
    with torch.cuda.stream(s):
        pool.batch_gate_dev(O.GATE_NAND, ta, ta, mid, stream=s)   # not(a)
        pool.batch_gate_dev(O.GATE_NAND, mid, mid, out, stream=s)  # not(not(a))
    s.synchronize()
    pool.synchronize()
    assert np.array_equal(sk.decrypt_bool(_host(out)), np.array([1, 0, 1, 1], bool))
    assert np.array_equal(_host(out), eng128.batch_gate(O.GATE_NAND, _host(mid), _host(mid)))
    pool.close()


def test_pool_host_forms_of_the_whole_batch_api(O, eng128, keys128):
    """tfhe_hip_pool_batch_{gates_mixed_nks, tlwe_lincomb, lincomb_bootstrap}: the host-pointer pool calls round 3 lacked."""
    import rs_tfhe_amd as R

    sk, ck = keys128
    pool = R.Pool(eng128.params, [0, 0])
    pool.load_cloud_key(_cloud_key(ck))
    rng = np.random.default_rng(9400)
    for count in (7, 600):
        ca = sk.encrypt_bool(rng.integers(0, 2, count).astype(bool), 9401 + count)
        cb = sk.encrypt_bool(rng.integers(0, 2, count).astype(bool), 9402 + count)
        codes = rng.integers(0, 11, count).astype(np.uint8)
        tv = rng.integers(0, 2**32, (count, 2, N), dtype=np.uint64).astype(np.uint32)
        assert np.array_equal(pool.batch_gates_mixed(codes, ca, cb, keyswitch=False), eng128.batch_gates_mixed(codes, ca, cb, keyswitch=False))
        assert np.array_equal(pool.batch_tlwe_lincomb(1, ca, -1, cb, 5), eng128.batch_tlwe_lincomb(1, ca, -1, cb, 5))
        assert np.array_equal(pool.batch_tlwe_lincomb(-1, ca), eng128.batch_tlwe_lincomb(-1, ca))
        assert np.array_equal(pool.batch_lincomb_bootstrap(1, ca, 2, cb, 0x40000000), eng128.batch_lincomb_bootstrap(1, ca, 2, cb, 0x40000000))
        assert np.array_equal(pool.batch_lincomb_bootstrap(1, ca, 1, cb, 0, testvec=tv, keyswitch=False),
                              eng128.batch_lincomb_bootstrap(1, ca, 1, cb, 0, testvec=tv, keyswitch=False))
    pool.close()


def test_configs4_level_and_adder_through_one_pool_handle(O, keys80, keys128, eng128):
    """BASELINE configs[4]'s circuit level (M Gates::mux in the reference's formula + X hom_xor: two blind-rotation
    launches and one key switch, gates.rs:157-183) and the ripple-carry adder (examples/add_two_numbers.rs) run
    through ONE pool handle -- `circuit.mux_and_gates_dev` / `Circuit.run` take a Pool wherever they take an Engine --
    and give the single engine's words."""
    import torch

    import rs_tfhe_amd as R
    from rs_tfhe_amd import circuit as Cq

    sk, ck = keys80
    pk = _cloud_key(ck)
    eng = R.bootstrap.engine_for(pk.params, 0)
    eng.ensure_key(pk)
    pool = R.Pool(pk.params, [0, 0])
    pool.load_cloud_key(pk)
    rng = np.random.default_rng(9500)
    M, X = 400, 333
    bits = [rng.integers(0, 2, M).astype(bool) for _ in range(3)] + [rng.integers(0, 2, X).astype(bool) for _ in range(2)]
    cts = [_dev(sk.encrypt_bool(b, 9501 + i)) for i, b in enumerate(bits)]
    codes = torch.full((X,), O.GATE_XOR, dtype=torch.uint8, device="cuda:0")
    for home in (0, 1):
        pool.home = home
        m1, x1 = Cq.mux_and_gates_dev(eng, *cts[:3], codes, *cts[3:])
        m2, x2 = Cq.mux_and_gates_dev(pool, *cts[:3], codes, *cts[3:])
        pool.synchronize()
        torch.cuda.synchronize()
        assert np.array_equal(_host(m2), _host(m1)) and np.array_equal(_host(x2), _host(x1))
        assert np.array_equal(sk.decrypt_bool(_host(x2)), bits[3] ^ bits[4])
    assert pool.data_transport == "peer-copy"
    pool.close()
    # the adder, 128-bit, 4 bits x 200 independent inputs: 9 levels, each one pool call cut over both members
    sk, ck = keys128
    pool = R.Pool(eng128.params, [0, 0])
    pool.load_cloud_key(_cloud_key(ck))
    nbits, B = 4, 200
    xs, ys = rng.integers(0, 16, B), rng.integers(0, 16, B)
    c = R.Circuit(2 * nbits + 1)
    sum_w, carry_w = c.add(list(range(nbits)), list(range(nbits, 2 * nbits)), 2 * nbits)
    bitsm = np.zeros((2 * nbits + 1, B), bool)
    for i in range(nbits):
        bitsm[i] = (xs >> i) & 1
        bitsm[nbits + i] = (ys >> i) & 1
    inputs = np.stack([sk.encrypt_bool(bitsm[w], 9600 + w) for w in range(2 * nbits + 1)])
    wires = c.run(pool, inputs)
    assert np.array_equal(wires, c.run(eng128, inputs))
    total = np.zeros(B, np.int64)
    for i, w in enumerate(sum_w):
        total += sk.decrypt_bool(wires[w]).astype(np.int64) << i
    total += sk.decrypt_bool(wires[carry_w]).astype(np.int64) << nbits
    assert np.array_equal(total, xs + ys)
    pool.close()


def test_pool_views_share_the_parents_lock_and_communicator(O, eng128, keys128, monkeypatch):
    """A key view of a pool runs on its parent's members, staging buffers, communicator and mutex: device-resident calls
    under two keys interleave correctly, and destroying the parent first keeps it alive until its views are gone."""
    import torch

    import rs_tfhe_amd as R

    sk1, ck1 = keys128
    sk2, ck2 = oracle_keys(O, O.SECURITY_128_BIT, seed=4321)
    monkeypatch.setenv("TFHE_HIP_POOL_RCCL", "2")
    pool = R.Pool(eng128.params, [0])
    pool.load_cloud_key(_cloud_key(ck1))
    view = pool.new_key_view()
    view.load_cloud_key(_cloud_key(ck2))
    A = np.array([1, 0, 0, 1, 1, 0, 1], bool)
    t1, t2 = _dev(sk1.encrypt_bool(A, 9701)), _dev(sk2.encrypt_bool(A, 9702))
    o1, o2 = torch.empty_like(t1), torch.empty_like(t2)
    for _ in range(3):
        pool.batch_gate_dev(O.GATE_NAND, t1, t1, o1)
        view.batch_gate_dev(O.GATE_NAND, t2, t2, o2)
    pool.synchronize()
    torch.cuda.synchronize()
    assert np.array_equal(sk1.decrypt_bool(_host(o1)), ~A) and np.array_equal(sk2.decrypt_bool(_host(o2)), ~A)
    assert view.data_transport == "rccl"
    # parent destroyed first through the C ABI: the view still works, and frees the parent when it goes
    lib = pool._lib
    h_parent, h_view = pool._h, view._h
    pool._h = None
    pool._views = []
    lib.tfhe_hip_pool_destroy(h_parent)
    view.batch_gate_dev(O.GATE_NAND, t2, t2, o2)
    view.synchronize()
    torch.cuda.synchronize()
    assert np.array_equal(sk2.decrypt_bool(_host(o2)), ~A)
    view._h = None
    view._parent = None
    lib.tfhe_hip_pool_destroy(h_view)


# ---- the 8-GPU workloads of BASELINE.json at their GLOBAL size, through the one handle a caller would use ---------------
def _spread_per_shard(pool, count, world, per_shard=256, edge=4):
    """Indices to hold against the CPU path: `per_shard` spread over every shard of the order-preserving cut plus the
    first and last `edge` of each (a wrong boundary, a swapped shard or a short transfer shows at the edges first)."""
    idx = []
    for r in range(world):
        lo, hi = pool.shard(count, r)
        idx.append(np.r_[lo:lo + edge, np.linspace(lo, hi - 1, per_shard).astype(np.int64), hi - edge:hi])
    return np.unique(np.concatenate(idx))


def _encrypt_chunks(sk, bits, seed, chunk=65536):
    """[len(bits)][n+1] fresh encryptions, generated chunk by chunk (distinct seeds) into one array."""
    out = np.empty((len(bits), sk.params.n + 1), np.uint32)
    for i, lo in enumerate(range(0, len(bits), chunk)):
        out[lo:lo + chunk] = sk.encrypt_bool(bits[lo:lo + chunk], seed + i)
    return out


def _decrypt_all(sk, out, chunk=131072):
    n = sk.params.n
    key = sk.key_lv0.astype(np.uint32)[None, :]
    res = np.empty(len(out), bool)
    for lo in range(0, len(out), chunk):
        o = out[lo:lo + chunk]
        res[lo:lo + chunk] = (o[:, n] - (o[:, :n] * key).sum(axis=1, dtype=np.uint32)).view(np.int32) >= 0
    return res


def test_configs2_global_batch_through_eight_member_pool(O, eng128, keys128):
    """BASELINE configs[2] at its GLOBAL size: 524,288 hom_nand at SECURITY_128_BIT on 524,288 distinct ciphertext
    pairs, resident on member 0's GPU, through ONE `tfhe_hip_pool_batch_gate_dev` call on an eight-member pool
    (devices = [0] * 8: the box has one GPU, so the seven remote shards travel by peer copies -- the staging buffers,
    shard arithmetic, events and order-preserving gather are the ones eight GPUs run; only the physical xGMI links are
    not exercised).  Bar: every output decrypts to nand(a, b); >= 256 outputs spread over EVERY shard plus each shard's
    first and last four equal the CPU path word for word (gates.rs:357-383: results in input order)."""
    import torch

    import rs_tfhe_amd as R

    sk, ck = keys128
    pool = R.Pool(eng128.params, [0] * 8)
    pool.load_cloud_key(_cloud_key(ck))
    B = 524288
    assert pool.members_for(B) == 8
    rng = np.random.default_rng(9800)
    bits_a, bits_b = rng.integers(0, 2, B).astype(bool), rng.integers(0, 2, B).astype(bool)
    ca, cb = _encrypt_chunks(sk, bits_a, 9801), _encrypt_chunks(sk, bits_b, 9901)
    ta, tb = _dev(ca), _dev(cb)
    to = torch.full_like(ta, 0x5A5A5A5A)
    pool.set_profiling(True)
    pool.transfer_times()
    pool.batch_gate_dev(O.GATE_NAND, ta, tb, to, home=0)
    pool.synchronize()
    torch.cuda.synchronize()
    tt = pool.transfer_times()
    pool.set_profiling(False)
    assert pool.data_transport == "peer-copy"
    lo0, hi0 = pool.shard(B, 0)
    moved = B - (hi0 - lo0)
    assert tt["calls"] == 1 and tt["scatter_bytes"] == 2 * moved * 701 * 4 and tt["gather_bytes"] == moved * 701 * 4
    out = _host(to)
    del ta, tb
    assert np.array_equal(_decrypt_all(sk, out), ~(bits_a & bits_b))
    idx = _spread_per_shard(pool, B, 8)
    assert len(idx) >= 8 * 256
    assert np.array_equal(out[idx], O.batch_gate(ck, O.GATE_NAND, ca[idx], cb[idx]))
    pool.close()


def test_every_home_member_of_an_eight_member_pool(O, eng128, keys128):
    """Shard r runs on member (home + r) mod 8: every member 0..7 as HOME once, on a ragged batch cut eight ways (and on
    one cut three ways), each against the single context word for word; the first and last result of every shard also
    against the CPU path."""
    import torch

    import rs_tfhe_amd as R

    sk, ck = keys128
    pool = R.Pool(eng128.params, [0] * 8)
    pool.load_cloud_key(_cloud_key(ck))
    rng = np.random.default_rng(9850)
    for count in (8 * 300 + 5, 700):
        A, B = rng.integers(0, 2, count).astype(bool), rng.integers(0, 2, count).astype(bool)
        ca, cb = sk.encrypt_bool(A, 9851 + count), sk.encrypt_bool(B, 9852 + count)
        ta, tb = _dev(ca), _dev(cb)
        want = torch.empty_like(ta)
        eng128.batch_gate_dev(O.GATE_NAND, ta, tb, want)
        torch.cuda.synchronize()
        w = _host(want)
        world = pool.members_for(count)
        assert world == (8 if count > 2048 else 3)
        edges = np.unique(np.concatenate([np.r_[pool.shard(count, r)[0], pool.shard(count, r)[1] - 1] for r in range(world)]))
        assert np.array_equal(w[edges], O.batch_gate(ck, O.GATE_NAND, ca[edges], cb[edges]))
        for home in range(8):
            got = torch.full_like(ta, 0x5A5A5A5A)
            pool.batch_gate_dev(O.GATE_NAND, ta, tb, got, home=home)
            pool.synchronize()
            torch.cuda.synchronize()
            assert np.array_equal(_host(got), w), (count, home)
    pool.close()


def test_configs4_global_batch_through_eight_member_pool(O, keys80):
    """BASELINE configs[4] at its GLOBAL size: 2^20 gates at SECURITY_80_BIT, half `Gates::mux` (the reference's formula,
    gates.rs:157-183, quirk Q5) and half hom_xor, all 5 x 524,288 input ciphertexts distinct and resident on member 0's
    GPU, as ONE circuit level through ONE eight-member pool handle (`circuit.mux_and_gates_dev`: two pool calls of 2^20
    items each, cut eight ways).  Bar: the xor half decrypts; >= 256 mux and >= 256 xor outputs spread over every shard of
    the second call's cut (plus shard edges) equal the CPU path word for word."""
    import torch

    import rs_tfhe_amd as R

    sk, ck = keys80
    pk = _cloud_key(ck)
    pool = R.Pool(pk.params, [0] * 8)
    pool.load_cloud_key(pk)
    M = X = 524288
    rng = np.random.default_rng(9900)
    bits = rng.integers(0, 2, (5, M)).astype(bool)
    a, b, c, xa, xb = (_encrypt_chunks(sk, bits[w], 10000 + 100 * w) for w in range(5))
    ta, tb, tc, txa, txb = (_dev(x) for x in (a, b, c, xa, xb))
    codes = torch.full((X,), R.engine.XOR, dtype=torch.uint8, device="cuda:0")
    mo, xo = R.circuit.mux_and_gates_dev(pool, ta, tb, tc, codes, txa, txb)
    pool.synchronize()
    torch.cuda.synchronize()
    assert pool.data_transport == "peer-copy"
    mux, xor = _host(mo).copy(), _host(xo).copy()
    del ta, tb, tc, txa, txb, mo, xo
    assert np.array_equal(_decrypt_all(sk, xor), bits[3] ^ bits[4])
    # the second pool call holds [M mux | X xor] as ONE batch of M + X items: its eight shards are halves 0-3 (mux) and
    # 4-7 (xor); take the spread over that cut
    idx = _spread_per_shard(pool, M + X, 8)
    im, ix = idx[idx < M], idx[idx >= M] - M
    assert len(im) >= 4 * 256 and len(ix) >= 4 * 256
    assert np.array_equal(mux[im], O.batch_mux(ck, a[im], b[im], c[im], naive=False))
    assert np.array_equal(xor[ix], O.batch_gate(ck, O.GATE_XOR, xa[ix], xb[ix]))
    pool.close()
