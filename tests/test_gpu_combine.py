"""The combining front end (rs-tfhe_amd/csrc/combine.hpp): concurrent SMALL host-pointer calls on one handle share launches.

The reference's strategies are `Send + Sync` (src/bootstrap/mod.rs:23-38) and SURVEY section 8(b) names the usage: "`&self`
may be called concurrently from Rayon workers, e.g. user code doing `par_iter` over gates".  A team of C++ host threads
(rs-tfhe_amd/csrc/callers.cpp) makes one-ciphertext calls through the C ABI; every returned word is held to the CPU
checker, and the aggregate rate is held against the same calls made one at a time (front end switched off).
"""
import threading

import numpy as np
import pytest

from test_gpu_parity import _cloud_key, eng128  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu
N = 1024


def _oracle_gates(O, ck, gates, ca, cb):
    """per-call gate codes -> the CPU path's outputs, one orc_batch_gate per distinct gate"""
    exp = np.zeros_like(ca)
    for g in np.unique(gates):
        idx = np.nonzero(gates == g)[0]
        exp[idx] = O.batch_gate(ck, int(g), ca[idx], cb[idx])
    return exp


def test_sixty_four_threads_of_single_gates_share_launches(O, eng128, keys128):
    """64 threads x 50 one-ciphertext calls of `Gates::nand` / `xor` / ... on ONE context: every result equals the CPU
    path word for word, the calls were merged (tens of calls per launch), and the team gets at least 20 x what the same
    calls get one at a time (measured on MI355X: ~26 k gates/s against ~460)."""
    from rs_tfhe_amd import callers

    sk, ck = keys128
    T, K = 64, 50
    rng = np.random.default_rng(601)
    A, B = rng.integers(0, 2, T * K).astype(bool), rng.integers(0, 2, T * K).astype(bool)
    gates = rng.choice(np.array([O.GATE_NAND, O.GATE_XOR, O.GATE_AND, O.GATE_OR, O.GATE_XNOR, O.GATE_ANDNY], np.uint8), T * K)
    ca, cb = sk.encrypt_bool(A, 6001), sk.encrypt_bool(B, 6002)
    eng128.combine_stats()
    out, secs, call_ms = callers.run(eng128, callers.OP_GATE, ca, cb, gates=gates, threads=T, calls=K)
    st = eng128.combine_stats()
    assert np.array_equal(out, _oracle_gates(O, ck, gates, ca, cb))
    assert st["requests"] == T * K and st["ciphertexts"] == T * K
    assert st["launches"] * 8 <= st["requests"] and st["max_requests_per_launch"] >= T // 2, st
    merged_rate = T * K / secs
    # the same calls one at a time: front end off, every call takes the context's mutex for its whole duration
    bound = st["max_count"]
    eng128.set_combining(0)
    try:
        T1, K1 = 8, 12
        out1, secs1, _ = callers.run(eng128, callers.OP_GATE, ca[: T1 * K1], cb[: T1 * K1], gates=gates[: T1 * K1], threads=T1, calls=K1)
    finally:
        eng128.set_combining(bound)
    k = np.arange(T1 * K1)
    assert np.array_equal(out1, out[k])  # merged or not: the same bits
    serial_rate = T1 * K1 / secs1
    assert merged_rate >= 20 * serial_rate, (merged_rate, serial_rate, st)
    print(f"merged {merged_rate:.0f} gates/s, one at a time {serial_rate:.0f} gates/s, {st}")


def test_concurrent_lut_bootstraps_with_their_own_tables(O, eng128, keys128):
    """LutBootstrap::bootstrap_lut (lut.rs:79-99) from 32 threads, every call with its OWN lookup table: the merged
    launch carries per-ciphertext test vectors."""
    from rs_tfhe_amd import callers

    sk, ck = keys128
    T, K = 32, 6
    rng = np.random.default_rng(602)
    msgs = rng.integers(0, 2, T * K)
    cts = sk.encrypt_lwe_message(msgs, 2, 6003)
    tables = [O.lut_generate(f, 2) for f in (lambda x: x, lambda x: 1 - x, lambda x: 1, lambda x: 0)]
    which = rng.integers(0, 4, T * K)
    tvs = np.stack([tables[w] for w in which])
    out, _, _ = callers.run(eng128, callers.OP_BOOTSTRAP_LUT, cts, testvecs=tvs, threads=T, calls=K)
    assert np.array_equal(out, O.batch_bootstrap(ck, cts, testvec=tvs))
    fs = (lambda x: x, lambda x: 1 - x, lambda x: 1, lambda x: 0)
    assert np.array_equal(sk.decrypt_lwe_message(out, 2), np.array([fs[w](int(m)) % 2 for w, m in zip(which, msgs)]))


def test_concurrent_single_gates_on_a_pool(O, eng128, keys128):
    """The same team on a pool handle (two members on this box's GPU): small calls skip the pool's mutex and merge in the
    front end of the least loaded member -- members that share a device count once, so here the first member takes
    them all (tests/test_gpu_multi_device.py has the spread over distinct devices)."""
    import rs_tfhe_amd as R
    from rs_tfhe_amd import callers

    sk, ck = keys128
    pk = _cloud_key(ck)
    pool = R.Pool(pk.params, [0, 0])
    pool.load_cloud_key(pk)
    T, K = 64, 20
    rng = np.random.default_rng(603)
    A, B = rng.integers(0, 2, T * K).astype(bool), rng.integers(0, 2, T * K).astype(bool)
    gates = rng.choice(np.array([O.GATE_NAND, O.GATE_XOR, O.GATE_NOR], np.uint8), T * K)
    ca, cb = sk.encrypt_bool(A, 6004), sk.encrypt_bool(B, 6005)
    pool.combine_stats()
    out, secs, _ = callers.run(pool, callers.OP_GATE, ca, cb, gates=gates, threads=T, calls=K)
    st = pool.combine_stats()
    assert np.array_equal(out, _oracle_gates(O, ck, gates, ca, cb))
    assert st[0]["requests"] == T * K and st[1]["requests"] == 0, st
    assert st[0]["launches"] * 8 <= T * K, st
    # mux through the pool, merged as well (three bootstraps each)
    Cc = rng.integers(0, 2, 48).astype(bool)
    cc = sk.encrypt_bool(Cc, 6006)
    naive = (np.arange(48) % 2).astype(np.uint8)
    outm, _, _ = callers.run(pool, callers.OP_MUX, ca[:48], cb[:48], cc, gates=naive, threads=16, calls=3)
    for flag in (0, 1):
        idx = np.nonzero(naive == flag)[0]
        assert np.array_equal(outm[idx], O.batch_mux(ck, ca[idx], cb[idx], cc[idx], naive=bool(flag)))
    # a key view of the pool (a second cloud key on every member): small calls under it merge on the same front end, in
    # launches of their own
    sk2, ck2 = O.keygen(O.SECURITY_128_BIT, 4321)
    pv = pool.new_key_view()
    pv.load_cloud_key(_cloud_key(ck2))
    a2, b2 = sk2.encrypt_bool(A[:64], 6007), sk2.encrypt_bool(B[:64], 6008)
    out2, _, _ = callers.run(pv, callers.OP_GATE, a2, b2, gates=gates[:64], threads=16, calls=4)
    assert np.array_equal(out2, _oracle_gates(O, ck2, gates[:64], a2, b2))
    out1, _, _ = callers.run(pool, callers.OP_GATE, ca[:64], cb[:64], gates=gates[:64], threads=16, calls=4)
    assert np.array_equal(out1, out[:64])  # the pool's own key is still the one its calls run under
    pv.close()
    pool.close()
    print(f"pool: {T * K / secs:.0f} gates/s, {st}")


def test_a_lone_caller_leads_at_once(O, eng128, keys128):
    """One thread, one call at a time: every call is its own launch, nobody lingers, and the call costs what a
    one-ciphertext call costs (2.2 ms of kernels)."""
    from rs_tfhe_amd import callers

    sk, ck = keys128
    A = np.array([1, 0, 1, 1, 0, 0, 1, 0] * 4, bool)
    ca, cb = sk.encrypt_bool(A, 6007), sk.encrypt_bool(~A, 6008)
    eng128.combine_stats()
    out, secs, call_ms = callers.run(eng128, callers.OP_GATE, ca, cb, gates=np.full(32, O.GATE_NAND, np.uint8), threads=1, calls=32)
    st = eng128.combine_stats()
    assert np.array_equal(out, O.batch_gate(ck, O.GATE_NAND, ca, cb))
    assert st["launches"] == 32 and st["requests"] == 32 and st["max_requests_per_launch"] == 1 and st["lingers"] == 0, st
    print(f"lone caller: median {np.median(call_ms):.2f} ms per call")


def test_every_kind_of_small_call_at_once_under_two_keys(O, eng128, keys128):
    """Gates, mixed gates, bootstraps with and without key switch, with shared and per-ciphertext tables, mux and
    mux_naive, at ragged counts, from Python threads on TWO key views of one context at once: the leader sorts what it
    takes into one launch per (key view, operation class); every caller gets its own rows."""
    sk, ck = keys128
    sk2, ck2 = O.keygen(O.SECURITY_128_BIT, 4321)
    v2 = eng128.new_key_view()
    v2.load_cloud_key(_cloud_key(ck2))
    rng = np.random.default_rng(604)
    jobs = []
    for t in range(33):
        eng, s, k = (eng128, sk, ck) if t % 2 == 0 else (v2, sk2, ck2)
        n = int(rng.integers(1, 9))
        a, b, c = (s.encrypt_bool(rng.integers(0, 2, n).astype(bool), 7000 + 3 * t + j) for j in range(3))
        kind = t % 11
        if kind == 0:
            jobs.append((lambda e=eng, a=a, b=b: e.batch_gate(O.GATE_XOR, a, b), lambda k=k, a=a, b=b: O.batch_gate(k, O.GATE_XOR, a, b)))
        elif kind == 1:
            codes = rng.integers(0, 11, n).astype(np.uint8)
            jobs.append((lambda e=eng, a=a, b=b, codes=codes: e.batch_gates_mixed(codes, a, b),
                         lambda k=k, a=a, b=b, codes=codes: _oracle_gates(O, k, codes, a, b)))
        elif kind == 2:
            jobs.append((lambda e=eng, a=a: e.batch_bootstrap(a), lambda k=k, a=a: O.batch_bootstrap(k, a)))
        elif kind == 3:
            jobs.append((lambda e=eng, a=a: e.batch_bootstrap(a, keyswitch=False), lambda k=k, a=a: O.batch_bootstrap(k, a, keyswitch=False)))
        elif kind == 4:
            tv = rng.integers(0, 2**32, (2, N), dtype=np.uint64).astype(np.uint32)
            jobs.append((lambda e=eng, a=a, tv=tv: e.batch_bootstrap(a, tv), lambda k=k, a=a, tv=tv: O.batch_bootstrap(k, a, testvec=tv)))
        elif kind == 5:
            tv = rng.integers(0, 2**32, (n, 2, N), dtype=np.uint64).astype(np.uint32)
            jobs.append((lambda e=eng, a=a, tv=tv: e.batch_bootstrap(a, tv, keyswitch=False),
                         lambda k=k, a=a, tv=tv: O.batch_bootstrap(k, a, testvec=tv, keyswitch=False)))
        elif kind == 6:
            jobs.append((lambda e=eng, a=a, b=b, c=c: e.batch_mux(a, b, c, naive=False), lambda k=k, a=a, b=b, c=c: O.batch_mux(k, a, b, c, naive=False)))
        elif kind == 7:
            jobs.append((lambda e=eng, a=a, b=b, c=c: e.batch_mux(a, b, c, naive=True), lambda k=k, a=a, b=b, c=c: O.batch_mux(k, a, b, c, naive=True)))
        elif kind == 9:  # trgsw::blind_rotate: the output is the TRLWE
            jobs.append((lambda e=eng, a=a: e.batch_blind_rotate(a), lambda k=k, a=a: O.batch_blind_rotate(k, a)))
        elif kind == 10:
            tv = rng.integers(0, 2**32, (2, N), dtype=np.uint64).astype(np.uint32)
            jobs.append((lambda e=eng, a=a, tv=tv: e.batch_blind_rotate(a, tv), lambda k=k, a=a, tv=tv: O.batch_blind_rotate(k, a, testvec=tv)))
        else:  # a linear combination bootstrapped through a table (tfhe_hip_batch_lincomb_bootstrap: formed by the caller, merged as COPY)
            tv = rng.integers(0, 2**32, (2, N), dtype=np.uint64).astype(np.uint32)
            prep = (np.uint32(3) * a + np.uint32(0xFFFFFFFE) * b).astype(np.uint32)
            prep[:, -1] += np.uint32(0x10000000)
            jobs.append((lambda e=eng, a=a, b=b, tv=tv: e.batch_lincomb_bootstrap(3, a, 0xFFFFFFFE, b, 0x10000000, testvec=tv),
                         lambda k=k, prep=prep, tv=tv: O.batch_bootstrap(k, prep, testvec=tv)))
    results, errors = [None] * len(jobs), []

    def work(i):
        try:
            for _ in range(3):
                results[i] = jobs[i][0]()
        except Exception as e:  # noqa: BLE001
            errors.append((i, e))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for i, (_, ref) in enumerate(jobs):
        assert np.array_equal(results[i], ref()), i
    v2.close()


def test_errors_stay_with_the_thread_that_caused_them(O, eng128, keys128):
    """Threads that make bad calls (unknown gate, a view without a key) beside threads that make good ones: every bad
    call fails with its own text in its own thread, every good call is right."""
    import rs_tfhe_amd as R

    sk, ck = keys128
    empty = eng128.new_key_view()  # no key loaded
    A = np.array([1, 0, 1], bool)
    ca, cb = sk.encrypt_bool(A, 6101), sk.encrypt_bool(~A, 6102)
    exp = O.batch_gate(ck, O.GATE_NAND, ca, cb)
    seen = {}

    def good(i):
        for _ in range(10):
            if not np.array_equal(eng128.batch_gate(O.GATE_NAND, ca, cb), exp):
                seen[i] = "wrong words"
                return
        seen[i] = "ok"

    def bad_gate(i):
        for _ in range(10):
            try:
                eng128.batch_gate(77, ca, cb)
                seen[i] = "no error"
                return
            except R._capi.TfheHipError as e:
                if e.code != R._capi.EINVAL or "unknown gate" not in str(e):
                    seen[i] = str(e)
                    return
        seen[i] = "ok"

    def no_key(i):
        for _ in range(10):
            try:
                empty.batch_gate(O.GATE_NAND, ca, cb)
                seen[i] = "no error"
                return
            except R._capi.TfheHipError as e:
                if e.code != R._capi.ENOKEY or "not loaded" not in str(e):
                    seen[i] = str(e)
                    return
        seen[i] = "ok"

    threads = [threading.Thread(target=f, args=(i,)) for i, f in enumerate([good, bad_gate, no_key] * 4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert all(v == "ok" for v in seen.values()) and len(seen) == 12, seen
    empty.close()


def test_small_calls_beside_a_large_one_and_a_key_reload(O, eng128, keys128):
    """A large batch on the context's own stream while small calls merge on the lanes, then a key reload: the reload
    waits for the lanes (comb_quiesce) and the next small calls run under the new key."""
    sk, ck = keys128
    sk2, ck2 = O.keygen(O.SECURITY_128_BIT, 4321)
    rng = np.random.default_rng(605)
    big_a = sk.encrypt_bool(rng.integers(0, 2, 1500).astype(bool), 6201)
    big_b = sk.encrypt_bool(rng.integers(0, 2, 1500).astype(bool), 6202)
    small_a, small_b = big_a[:5], big_b[:5]
    got = {}

    def big():
        got["big"] = eng128.batch_gate(O.GATE_NAND, big_a, big_b)

    def small(i):
        got[i] = [eng128.batch_gate(O.GATE_XOR, small_a, small_b) for _ in range(4)]

    threads = [threading.Thread(target=big)] + [threading.Thread(target=small, args=(i,)) for i in range(6)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    exp_small = O.batch_gate(ck, O.GATE_XOR, small_a, small_b)
    assert all(np.array_equal(x, exp_small) for i in range(6) for x in got[i])
    assert np.array_equal(got["big"][:64], O.batch_gate(ck, O.GATE_NAND, big_a[:64], big_b[:64]))
    assert np.array_equal(sk.decrypt_bool(got["big"]), ~(sk.decrypt_bool(big_a) & sk.decrypt_bool(big_b)))
    view = eng128.new_key_view()
    view.load_cloud_key(_cloud_key(ck))
    a1 = sk.encrypt_bool(np.array([1, 0], bool), 6203)
    assert np.array_equal(view.batch_gate(O.GATE_NAND, a1, a1), O.batch_gate(ck, O.GATE_NAND, a1, a1))
    view.load_cloud_key(_cloud_key(ck2))  # reload under the same handle
    a2 = sk2.encrypt_bool(np.array([1, 0], bool), 6204)
    assert np.array_equal(view.batch_gate(O.GATE_NAND, a2, a2), O.batch_gate(ck2, O.GATE_NAND, a2, a2))
    view.close()


@pytest.mark.parametrize("setname", ["SECURITY_UINT4", "SECURITY_80_BIT", "SECURITY_UINT1"])
def test_merged_calls_on_other_parameter_sets(O, setname):
    """The front end on the other instantiations: SECURITY_UINT4 (l = 1, general rounding, the base-32 key switch: per-call
    lookup tables at message modulus 16), SECURITY_80_BIT (n = 550: gates and both mux forms) and SECURITY_UINT1 (l = 2).  A
    merged launch runs the same kernels in the same operation order as one plain batch call of the same ciphertexts, so the
    bits must be those of the batch call (which the parity suite holds to the CPU path), whatever the rounding regime;
    and they decrypt."""
    import rs_tfhe_amd as R
    from conftest import oracle_keys
    from rs_tfhe_amd import callers

    sk, ck = oracle_keys(O, getattr(O, setname), with_time=(setname == "SECURITY_UINT4"))
    pk = _cloud_key(ck)
    eng = R.Engine(pk.params, 0)
    eng.load_cloud_key(pk)
    rng = np.random.default_rng(606)
    T, K = 24, 4
    n = T * K
    if setname == "SECURITY_UINT4":
        msgs = rng.integers(0, 16, n)
        cts = sk.encrypt_lwe_message(msgs, 16, 6301)
        fs = (lambda x: x % 16, lambda x: (x * x) % 16, lambda x: (15 - x) % 16)
        which = rng.integers(0, 3, n)
        tables = [R.lut.Generator(16).generate_lookup_table(f).poly for f in fs]
        tvs = np.stack([tables[w] for w in which])
        bound = eng.combine_stats()["max_count"]
        eng.set_combining(0)
        want = eng.batch_bootstrap(cts, tvs)  # one plain batch call, per-ciphertext tables
        eng.set_combining(bound)
        eng.combine_stats()
        out, _, _ = callers.run(eng, callers.OP_BOOTSTRAP_LUT, cts, testvecs=tvs, threads=T, calls=K)
        st = eng.combine_stats()
        assert np.array_equal(out, want)
        assert np.array_equal(sk.decrypt_lwe_message(out, 16), np.array([fs[w](int(m)) for w, m in zip(which, msgs)]))
    else:
        A, B, Cc = (rng.integers(0, 2, n).astype(bool) for _ in range(3))
        ca, cb, cc = sk.encrypt_bool(A, 6302), sk.encrypt_bool(B, 6303), sk.encrypt_bool(Cc, 6304)
        gates = rng.integers(0, 10, n).astype(np.uint8)
        naive = (np.arange(n) % 2).astype(np.uint8)
        bound = eng.combine_stats()["max_count"]
        eng.set_combining(0)
        want_g = eng.batch_gates_mixed(gates, ca, cb)
        want_m = np.where(naive[:, None] == 1, eng.batch_mux(ca, cb, cc, naive=True), eng.batch_mux(ca, cb, cc, naive=False))
        eng.set_combining(bound)
        assert np.array_equal(want_g[:24], np.stack([O.batch_gate(ck, int(g), ca[i:i + 1], cb[i:i + 1])[0] for i, g in enumerate(gates[:24])]))
        eng.combine_stats()
        out_g, _, _ = callers.run(eng, callers.OP_GATE, ca, cb, gates=gates, threads=T, calls=K)
        out_m, _, _ = callers.run(eng, callers.OP_MUX, ca, cb, cc, gates=naive, threads=T, calls=K)
        st = eng.combine_stats()
        assert np.array_equal(out_g, want_g) and np.array_equal(out_m, want_m)
        idx = np.nonzero(naive == 1)[0]
        assert np.array_equal(sk.decrypt_bool(out_m[idx]), np.where(A[idx], B[idx], Cc[idx]))
    assert st["launches"] * 3 <= st["requests"], st  # merged, not one by one
    eng.close()


def test_bulk_launches_yield_while_small_calls_arrive(O, eng128, keys128):
    """A merged launch needs whole CUs and cannot start while a launch of tens of thousands of ciphertexts holds them all
    (one-ciphertext calls beside 65,536-ciphertext batches: median 13 ms, p90 308 ms).  While small calls are arriving the
    batch kernel of a large call goes out in launches of 8,192 ciphertexts (include/tfhe_hip.h): the same words, more
    launches, and a small call waits for a chunk boundary.  Without small calls a batch is ONE launch."""
    import time

    import torch

    sk, ck = keys128
    rng = np.random.default_rng(607)
    B = 32768
    bits_a, bits_b = rng.integers(0, 2, B).astype(bool), rng.integers(0, 2, B).astype(bool)
    ca, cb = sk.encrypt_bool(bits_a[:4096], 6401), sk.encrypt_bool(bits_b[:4096], 6402)
    ca, cb = np.tile(ca, (8, 1)), np.tile(cb, (8, 1))
    ta, tb = (torch.from_numpy(x.view(np.int32)).cuda() for x in (ca, cb))
    quiet, busy = torch.empty_like(ta), torch.empty_like(ta)
    time.sleep(0.3)  # nothing small has arrived for 250 ms: one launch
    eng128.kernel_times()
    eng128.set_profiling(True)
    eng128.batch_gate_dev(O.GATE_NAND, ta, tb, quiet)
    eng128.synchronize()
    assert eng128.kernel_times()["blind_rotate_launches"] == 1
    want1 = O.batch_gate(ck, O.GATE_NAND, ca[:1], cb[:1])
    stop = threading.Event()
    lat = []

    def small():
        while not stop.is_set():
            t0 = time.perf_counter()
            out = eng128.batch_gate(O.GATE_NAND, ca[:1], cb[:1])
            lat.append((time.perf_counter() - t0) * 1e3)
            assert np.array_equal(out, want1)
            time.sleep(0.002)

    th = threading.Thread(target=small)
    th.start()
    time.sleep(0.05)
    eng128.kernel_times()
    for _ in range(4):
        eng128.batch_gate_dev(O.GATE_NAND, ta, tb, busy)
        eng128.synchronize()
    kt = eng128.kernel_times()
    stop.set()
    th.join()
    eng128.set_profiling(False)
    torch.cuda.synchronize()
    assert np.array_equal(busy.cpu().numpy(), quiet.cpu().numpy())  # chunked or not: the same words
    assert np.array_equal(quiet.cpu().numpy().view(np.uint32)[:64], O.batch_gate(ck, O.GATE_NAND, ca[:64], cb[:64]))
    # 4 batches of 32,768 in chunks of 8,192 = 16 batch-kernel launches (+ the small calls' own one-workgroup launches)
    assert kt["blind_rotate_launches"] >= 16, kt
    print(f"small calls beside chunked 32,768-ciphertext batches: {len(lat)} calls, median {np.median(lat):.1f} ms, max {max(lat):.1f} ms")


def test_rounds_larger_than_one_launch_are_cut(O, eng128, keys128):
    """A leader may take more than one launch holds (4,096 ciphertexts, the size of the lanes' arenas): 40 threads x 200
    gates = 8,000 in flight, and -- with the bound raised to its maximum -- three calls of 3,000 at once.  The round is cut
    into launches of whole requests; every caller still gets its own rows."""
    from rs_tfhe_amd import callers

    sk, ck = keys128
    rng = np.random.default_rng(608)
    T, K, per = 40, 2, 200
    n = T * K * per
    base_a = sk.encrypt_bool(rng.integers(0, 2, 2000).astype(bool), 6501)
    base_b = sk.encrypt_bool(rng.integers(0, 2, 2000).astype(bool), 6502)
    ca, cb = np.tile(base_a, (n // 2000, 1)), np.tile(base_b, (n // 2000, 1))
    gates = rng.choice(np.array([O.GATE_NAND, O.GATE_XOR], np.uint8), T * K)
    want = {int(g): eng128.batch_gates_mixed(np.full(2000, g, np.uint8), base_a, base_b) for g in np.unique(gates)}
    assert np.array_equal(want[O.GATE_NAND][:48], O.batch_gate(ck, O.GATE_NAND, base_a[:48], base_b[:48]))
    eng128.combine_stats()
    out, _, _ = callers.run(eng128, callers.OP_GATE, ca, cb, gates=gates, threads=T, calls=K, per_call=per)
    st = eng128.combine_stats()
    for k in range(T * K):
        rows = np.arange(k * per, (k + 1) * per)
        assert np.array_equal(out[rows], want[int(gates[k])][rows % 2000]), k
    assert st["requests"] == T * K and st["ciphertexts"] == n and st["launches"] >= 2
    eng128.set_combining(4096)
    try:
        res = [None] * 3

        def work(i):
            res[i] = eng128.batch_gate(O.GATE_XOR, ca[i * 3000:(i + 1) * 3000], cb[i * 3000:(i + 1) * 3000])

        ths = [threading.Thread(target=work, args=(i,)) for i in range(3)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        for i in range(3):
            rows = np.arange(i * 3000, (i + 1) * 3000)
            assert np.array_equal(res[i], want[O.GATE_XOR][rows % 2000]), i
    finally:
        eng128.set_combining(st["max_count"])
