"""GPU parity tests: the HIP path, called through the C ABI (libtfhe_hip.so via
rs_tfhe_amd), against the CPU oracle on the same seeded inputs and against the
committed golden fixtures.

Bars:
  * integer stages (gate prep, rotation, decomposition, sample extract, key switch):
    bit-exact;
  * FFT external product / whole blind rotation at bgbit=6 (128/80-bit): 0 LSB vs
    the exact-integer oracle (tolerance written as <= 1, asserted == 0 where measured);
  * SECURITY_UINT4 (bgbit=22): |diff| <= 2^9 LSB per external product vs the exact
    product (the reference's own f64 error class), decrypted messages identical.
"""
import numpy as np
import pytest

from conftest import oracle_keys, signed_diff

pytestmark = pytest.mark.gpu
N = 1024


def _product_params(op):
    from rs_tfhe_amd import params as P

    return P.PARAM_SETS[op.name]


def _cloud_key(ck):
    """oracle key material -> the product package's CloudKey (fields of src/key.rs:51-56)."""
    import rs_tfhe_amd as R

    if not hasattr(ck, "_product"):
        from rs_tfhe_amd.params import SecurityParams

        op = ck.params
        try:
            pp = _product_params(op)
        except KeyError:
            pp = SecurityParams(op.name, 0, op.n, op.l, op.bgbit, op.basebit, op.t, op.alpha_lv0, op.alpha_lv1)
        ck._product = R.CloudKey(pp, ck.bootstrapping_key, ck.key_switching_key, ck.decomposition_offset,
                                 ck.blind_rotate_testvec)
    return ck._product


@pytest.fixture
def eng128(O, keys128):
    """The process-wide engine for SECURITY_128_BIT with the oracle-generated key loaded.  Function
    scope on purpose: other tests load other keys into the same shared engine (ensure_key is a no-op
    when the key is already the loaded one)."""
    import rs_tfhe_amd as R

    sk, ck = keys128
    pk = _cloud_key(ck)
    eng = R.bootstrap.engine_for(pk.params, 0)
    eng.ensure_key(pk)
    return eng


# ---- FFT layer -------------------------------------------------------------------------
def test_ifft_matches_klemsa_layout(O, eng128):
    rng = np.random.default_rng(21)
    polys = rng.integers(0, 2**32, (8, N), dtype=np.uint64).astype(np.uint32)
    got = eng128.batch_ifft(polys)
    for p, g in zip(polys, got):
        ref = O.klemsa_ifft(p)
        assert np.abs(g - ref).max() <= 1e-11 * np.abs(ref).max()  # f64 round-off only


def test_fft_roundtrip_and_kats(O, eng128, golden):
    g = golden["stage"]
    rng = np.random.default_rng(22)
    polys = np.concatenate(
        [rng.integers(0, 2**32, (6, N), dtype=np.uint64).astype(np.uint32),
         g["kat_klemsa_roundtrip"][None], g["kat_delta"][None]])
    back = eng128.batch_fft(eng128.batch_ifft(polys))
    assert signed_diff(back, polys) == 0  # reference bound is < 2 (fft/mod.rs:119-133)
    # and through the oracle's spectra (cross-implementation)
    spectra = np.stack([O.klemsa_ifft(p) for p in polys])
    assert signed_diff(eng128.batch_fft(spectra), polys) == 0


def test_poly_mul_vs_schoolbook(O, eng128, golden):
    g = golden["stage"]
    rng = np.random.default_rng(23)
    a = [g["kat_consistency_a"], g["kat_dense_a"]]
    b = [g["kat_consistency_b"], g["kat_dense_b"]]
    for _ in range(14):
        a.append(rng.integers(0, 2**32, N, dtype=np.uint64).astype(np.uint32))
        b.append(rng.integers(0, 64, N, dtype=np.uint64).astype(np.uint32))
    a, b = np.stack(a), np.stack(b)
    got = eng128.batch_poly_mul(a, b)
    exp = np.stack([O.negacyclic_schoolbook(x, y) for x, y in zip(a, b)])
    assert signed_diff(got, exp) == 0  # reference bound < 2 (fft/mod.rs:136-159)
    assert np.array_equal(got[:2], np.stack([g["kat_consistency_expected"], g["kat_dense_expected"]]))


def test_stage_fft_rounds_half_away_from_zero(O, eng128):
    """FFTProcessor::fft ends in f64::round (klemsa.rs:145-146): exact .5 ties go AWAY from zero.  A constant
    spectrum x * ifft(delta_0) (and x * ifft(delta_512), the imaginary slot of the fold) comes back as exactly
    x at one coefficient and exactly 0 elsewhere -- every intermediate is exact -- so x = k + 1/2 lands on a
    tie.  tfhe_hip_batch_fft must agree with the reference semantics word for word (ties-to-even would give
    0, 2, 2, 0, -2, -2 on the first six)."""
    xs = np.array([0.5, 1.5, 2.5, -0.5, -1.5, -2.5, 1e6 + 0.5, -1e6 - 0.5, 2.0**31 - 0.5, -(2.0**31) + 0.5, 3.0, -7.0])
    want0 = [1, 2, 3, -1, -2, -3, 1000001, -1000001, -2**31, -2**31, 3, -7]
    for pos in (0, N // 2):
        delta = np.zeros(N, np.uint32)
        delta[pos] = 1
        unit = O.klemsa_ifft(delta)
        spectra = xs[:, None] * unit[None, :]
        ref = np.stack([O.klemsa_fft(sp) for sp in spectra])
        assert ref[:, pos].view(np.int32).tolist() == want0 and not np.delete(ref, pos, axis=1).any()
        got = eng128.batch_fft(spectra)
        assert np.array_equal(got, ref), (pos, got[:, pos].view(np.int32).tolist())


# ---- single stages ------------------------------------------------------------------------
def test_external_product_exact_128(O, eng128, keys128):
    sk, ck = keys128
    P = ck.params
    rng = np.random.default_rng(24)
    idx = np.array([0, 1, 2, 349, 350, 698, 699, 5], np.int32)
    t = rng.integers(0, 2**32, (len(idx), 2, N), dtype=np.uint64).astype(np.uint32)
    t[0] = 0
    t[1] = 0xFFFFFFFF
    got = eng128.batch_external_product(t, idx)
    for i, x, gg in zip(idx, t, got):
        exact = O.external_product_exact(ck.bootstrapping_key_time[i], x, P.l, P.bgbit, ck.decomposition_offset)
        assert signed_diff(gg, exact) == 0
        assert np.array_equal(gg, O.external_product_fft(ck.bootstrapping_key[i], x, P.l, P.bgbit, ck.decomposition_offset))


def test_cmux_selects_and_matches(O, eng128, keys128):
    """trgsw::cmux (trgsw.rs:174-196) = in1 + ExtProd(cond, in2 - in1), composed from the external
    product stage: word-for-word against the CPU path, and it selects by the key bit (the reference's
    own cmux test, trgsw.rs:469-505: BSK[i] encrypts key_lv0[i])."""
    sk, ck = keys128
    P = ck.params
    rng = np.random.default_rng(125)
    idx = np.array([0, 7, 123, 350, 699], np.int32)
    # in1 / in2: trivial TRLWE samples (a = 0) carrying +1/8 and -1/8 in every coefficient
    in1 = np.zeros((len(idx), 2, N), np.uint32)
    in2 = np.zeros((len(idx), 2, N), np.uint32)
    in1[:, 1, :] = 0x20000000
    in2[:, 1, :] = 0xE0000000
    got = in1 + eng128.batch_external_product(in2 - in1, idx)
    for i, x1, x2, g in zip(idx, in1, in2, got):
        assert np.array_equal(g, O.cmux(x1, x2, ck.bootstrapping_key[i], P.l, P.bgbit, ck.decomposition_offset))
        picked_second = bool(sk.key_lv0[i])  # cond = 1 selects in2
        phase = sk.trlwe_phase(g).view(np.int32)
        assert ((phase < 0) == picked_second).all()
    # random (non-trivial) operands: still word for word
    r1 = rng.integers(0, 2**32, (len(idx), 2, N), dtype=np.uint64).astype(np.uint32)
    r2 = rng.integers(0, 2**32, (len(idx), 2, N), dtype=np.uint64).astype(np.uint32)
    got = r1 + eng128.batch_external_product(r2 - r1, idx)
    for i, x1, x2, g in zip(idx, r1, r2, got):
        assert np.array_equal(g, O.cmux(x1, x2, ck.bootstrapping_key[i], P.l, P.bgbit, ck.decomposition_offset))


def test_sample_extract_bit_exact(O, eng128):
    rng = np.random.default_rng(25)
    t = rng.integers(0, 2**32, (5, 2, N), dtype=np.uint64).astype(np.uint32)
    got = eng128.batch_sample_extract(t)
    assert np.array_equal(got, np.stack([O.sample_extract_index(x, 0) for x in t]))
    for k in range(1, N):  # every k, as the reference's own test does (trlwe.rs:190-230)
        assert np.array_equal(eng128.batch_sample_extract(t[:2], k), np.stack([O.sample_extract_index(x, k) for x in t[:2]])), k
    for k in (1, 2, 511, 512, 1022, 1023):
        assert np.array_equal(eng128.batch_sample_extract(t, k), np.stack([O.sample_extract_index(x, k) for x in t]))
    from rs_tfhe_amd import _capi

    with pytest.raises(_capi.TfheHipError):
        eng128.batch_sample_extract(t, 1024)


def test_key_switch_bit_exact(O, eng128, keys128):
    sk, ck = keys128
    rng = np.random.default_rng(26)
    for count in (1, 7, 8, 9, 19):  # ragged vs the 8-ciphertext key-switch group
        lv1 = rng.integers(0, 2**32, (count, N + 1), dtype=np.uint64).astype(np.uint32)
        lv1[0, :N] = 0  # all digits zero after the precision offset... (a_bar = offset only)
        got = eng128.batch_identity_key_switch(lv1)
        exp = np.stack([O.identity_key_switching(ck, x) for x in lv1])
        assert np.array_equal(got, exp)


@pytest.mark.parametrize("setname,kernel", [
    ("SECURITY_128_BIT", "auto"),      # default dispatch: split kernel below 64, matrix cores (K in chunks) from 64
    ("SECURITY_128_BIT", "b4"),        # k_key_switch_b4 (LDS ring, base 4) at every count
    ("SECURITY_128_BIT", "generic"),   # k_key_switch (buffer loads)
    ("SECURITY_128_BIT", "mfma"),      # k_key_switch_mfma<11> (int8 matrix cores), t = 9; K chunks picked per launch (16 ... 4 here)
    ("SECURITY_128_BIT", "split"),     # k_key_switch_split at every count
    ("SECURITY_110_BIT", "mfma"),      # k_key_switch_mfma<5>, t = 8, 5,5,5,5 tiles
    ("SECURITY_80_BIT", "mfma"),       # k_key_switch_mfma<5>, t = 7, 5,5,4,4 tiles
    ("SECURITY_UINT1", "mfma"),        # k_key_switch_mfma<6>, t = 8
    ("SECURITY_UINT4", "auto"),        # split below 384, k_key_switch_sliced2 (base 32) from there: sets per lane and K chunks per launch
    ("SECURITY_UINT4", "sliced"),      # ... at every count (64 / 32 / 16 K chunks at these counts)
    ("SECURITY_UINT4", "generic"),     # k_key_switch (generic) at base 32
    ("SECURITY_UINT4", "split"),
    ("SECURITY_UINT2", "sliced"),      # base 16
    ("SECURITY_UINT3", "sliced"),      # base 64 (ring of two pairs), t = 2
    ("SECURITY_UINT7", "sliced"),      # base 128 (one workgroup per CU), n = 1160
    ("SECURITY_UINT3", "generic"),     # k_key_switch (generic) at base 64
])
def test_key_switch_batch_kernels_bit_exact(O, monkeypatch, setname, kernel):
    """Every key-switch kernel, forced at every batch size with the supported selector TFHE_HIP_KS_KERNEL
    (include/tfhe_hip.h), at ragged counts that cross the 32-, 128- and 512-ciphertext workgroup boundaries, against
    identity_key_switching (trgsw.rs:332-360) word for word.  (The per-launch choices -- K chunks, accumulator sets --
    follow from the count: test_dispatch_crossovers_bit_exact walks them.)"""
    import rs_tfhe_amd as R

    op = getattr(O, setname)
    sk, ck = oracle_keys(O, op)
    pk = _cloud_key(ck)
    monkeypatch.setenv("TFHE_HIP_KS_KERNEL", kernel)
    eng = R.Engine(pk.params, 0)
    eng.load_cloud_key(pk)
    rng = np.random.default_rng(27)
    for count in (1, 33, 515, 1300):  # several workgroups of rows
        if kernel != "auto":
            assert f"key_switch={kernel}" in eng.describe_dispatch(count)
        lv1 = rng.integers(0, 2**32, (count, N + 1), dtype=np.uint64).astype(np.uint32)
        lv1[0, :N] = 0
        lv1[-1, :N] = 0xFFFFFFFF
        got = eng.batch_identity_key_switch(lv1)
        exp = O.batch_identity_key_switching(ck, lv1) if hasattr(O, "batch_identity_key_switching") else \
            np.stack([O.identity_key_switching(ck, x) for x in lv1])
        assert np.array_equal(got, exp), (setname, kernel, count)
    eng.close()


def test_kernel_selectors_are_validated(monkeypatch):
    """TFHE_HIP_BR_KERNEL / TFHE_HIP_KS_KERNEL (the supported controls, include/tfhe_hip.h): unknown values and
    kernels the parameter set cannot run fail context creation with EINVAL and a message, nothing is allocated."""
    import rs_tfhe_amd as R
    from rs_tfhe_amd import _capi

    monkeypatch.setenv("TFHE_HIP_KS_KERNEL", "mfma")
    with pytest.raises(_capi.TfheHipError, match="not available"):
        R.Engine(R.params.SECURITY_UINT4, 0)  # base 32: no matrix-core form
    monkeypatch.setenv("TFHE_HIP_KS_KERNEL", "b4")
    with pytest.raises(_capi.TfheHipError, match="not available"):
        R.Engine(R.params.SECURITY_UINT4, 0)
    monkeypatch.setenv("TFHE_HIP_KS_KERNEL", "nonsense")
    with pytest.raises(_capi.TfheHipError, match="TFHE_HIP_KS_KERNEL must be"):
        R.Engine(R.params.SECURITY_128_BIT, 0)
    monkeypatch.delenv("TFHE_HIP_KS_KERNEL")
    monkeypatch.setenv("TFHE_HIP_BR_KERNEL", "wide")
    with pytest.raises(_capi.TfheHipError, match="TFHE_HIP_BR_KERNEL must be"):
        R.Engine(R.params.SECURITY_128_BIT, 0)


def test_blind_rotate_bit_exact(O, eng128, keys128):
    sk, ck = keys128
    cts = sk.encrypt_bool(np.array([1, 0, 1, 1, 0], bool), 27)
    cts[3, 700] = 0            # b_tilda = 2N
    cts[4, 700] = 0xFFFFFFFF   # b_tilda = 0 via the non-wrapping add (trgsw.rs:202-203)
    cts[4, 0] = 0xFFFFFFFF     # a_tilda = 0 via the wrapping add (trgsw.rs:210-211)
    got = eng128.batch_blind_rotate(cts)
    exp = O.batch_blind_rotate(ck, cts)
    assert signed_diff(got, exp) <= 1
    assert np.array_equal(got, exp)
    assert np.array_equal(got[0], O.blind_rotate(ck, cts[0], exact=True))


# ---- gates --------------------------------------------------------------------------------
@pytest.mark.parametrize("op", range(10))
def test_gates_bit_exact_and_truth_table(O, eng128, keys128, op):
    sk, ck = keys128
    A = np.array([1, 1, 0, 0, 1], bool)
    B = np.array([1, 0, 1, 0, 1], bool)
    ca, cb = sk.encrypt_bool(A, 300 + op), sk.encrypt_bool(B, 400 + op)
    got = eng128.batch_gate(op, ca, cb)
    assert np.array_equal(got, O.batch_gate(ck, op, ca, cb))
    exp = np.array([O.GATE_TRUTH[op](bool(a), bool(b)) for a, b in zip(A, B)])
    assert np.array_equal(sk.decrypt_bool(got), exp)


def test_reference_api_mirror(O, keys128):
    """Same call shapes as the reference: gates::nand(a,b,&ck), gates::batch_nand(pairs,&ck),
    Bootstrap::bootstrap(&ct,&ck), LutBootstrap::bootstrap_func(&ct,f,m,&ck)."""
    import rs_tfhe_amd as R

    sk, ck = keys128
    pk = _cloud_key(ck)
    ct_t, ct_f = sk.encrypt_bool([True], 501)[0], sk.encrypt_bool([False], 502)[0]
    out = R.gates.nand(ct_t, ct_f, pk)
    assert out.shape == (701,) and sk.decrypt_bool(out)[0]
    assert not sk.decrypt_bool(R.gates.nand(ct_t, ct_t, pk))[0]
    assert np.array_equal(out, O.batch_gate(ck, O.GATE_NAND, ct_t, ct_f)[0])
    pairs_a = np.stack([ct_t, ct_t, ct_f, ct_f])
    pairs_b = np.stack([ct_t, ct_f, ct_t, ct_f])
    for fn, truth in ((R.gates.batch_nand, lambda a, b: not (a and b)), (R.gates.batch_and, lambda a, b: a and b),
                      (R.gates.batch_or, lambda a, b: a or b), (R.gates.batch_xor, lambda a, b: a != b),
                      (R.gates.batch_nor, lambda a, b: not (a or b))):
        res = sk.decrypt_bool(fn(pairs_a, pairs_b, pk))
        assert list(res) == [truth(a, b) for a, b in ((1, 1), (1, 0), (0, 1), (0, 0))]
    bs = R.HipBootstrap()
    assert bs.name() == "hip-gfx950"
    assert sk.decrypt_bool(bs.bootstrap(ct_t, pk))[0] and not sk.decrypt_bool(bs.bootstrap(ct_f, pk))[0]
    assert np.array_equal(bs.bootstrap_without_key_switch(ct_t, pk), O.batch_bootstrap(ck, ct_t, keyswitch=False)[0])
    # non-fused strategy (Gates::with_bootstrap, gates.rs:43-45): host prep, then the strategy's
    # bootstrap() -- for LutBootstrap that is the m=2 identity LUT (lut.rs:108-111)
    g = R.Gates.with_bootstrap(R.LutBootstrap())
    assert g.bootstrap_strategy() == "lut-hip-gfx950"
    want = O.batch_bootstrap(ck, O.gate_prep(O.GATE_AND, ct_t, ct_f, 700), testvec=O.lut_generate(lambda x: x, 2))[0]
    assert np.array_equal(g.and_(ct_t, ct_f, pk), want)


def test_mux_variants(O, eng128, keys128):
    sk, ck = keys128
    A = np.array([0, 0, 0, 0, 1, 1, 1, 1], bool)
    B = np.array([0, 0, 1, 1, 0, 0, 1, 1], bool)
    Cc = np.array([0, 1, 0, 1, 0, 1, 0, 1], bool)
    ca, cb, cc = sk.encrypt_bool(A, 31), sk.encrypt_bool(B, 32), sk.encrypt_bool(Cc, 33)
    naive = eng128.batch_mux(ca, cb, cc, naive=True)
    assert np.array_equal(naive, O.batch_mux(ck, ca, cb, cc, naive=True))
    assert np.array_equal(sk.decrypt_bool(naive), np.where(A, B, Cc))  # gates.rs:656-681
    # Gates::mux: the reference formula reproduced bit-for-bit (quirk Q5; no decrypt claim)
    ref_formula = eng128.batch_mux(ca[:3], cb[:3], cc[:3], naive=False)
    assert np.array_equal(ref_formula, O.batch_mux(ck, ca[:3], cb[:3], cc[:3], naive=False))


# ---- programmable bootstrap ---------------------------------------------------------------
def test_lut_bootstrap_binary(O, eng128, keys128):
    """bootstrap/lut.rs:142-254 (identity / NOT / constant, m = 2) through the mirror API."""
    import rs_tfhe_amd as R

    sk, ck = keys128
    pk = _cloud_key(ck)
    lb = R.LutBootstrap()
    msgs = np.array([0, 1, 1, 0])
    cts = sk.encrypt_lwe_message(msgs, 2, 50)
    for f in (lambda x: x, lambda x: 1 - x, lambda x: 1):
        out = lb.bootstrap_func(cts, f, 2, pk)
        assert np.array_equal(out, O.batch_bootstrap(ck, cts, testvec=O.lut_generate(f, 2)))
        assert np.array_equal(sk.decrypt_lwe_message(out, 2), np.array([f(int(m)) % 2 for m in msgs]))
    lut = R.lut.Generator(2).generate_lookup_table(lambda x: 1 - x)  # LUT reuse (lut.rs:222-254)
    one = lb.bootstrap_lut(cts[1], lut, pk)
    assert one.shape == (701,) and sk.decrypt_lwe_message(one, 2)[0] == 0
    # per-ciphertext test vectors
    tvs = np.stack([O.lut_generate(lambda x: x, 2), O.lut_generate(lambda x: 1 - x, 2)] * 2)
    out = eng128.batch_bootstrap(cts, tvs)
    exp = np.stack([O.batch_bootstrap(ck, c, testvec=t)[0] for c, t in zip(cts, tvs)])
    assert np.array_equal(out, exp)


def test_pbs_uint4(O, keys_uint4):
    """BASELINE config 4: LutBootstrap, SECURITY_UINT4, message modulus 16."""
    import rs_tfhe_amd as R

    sk, ck = keys_uint4
    P = ck.params
    pk = _cloud_key(ck)
    eng = R.bootstrap.engine_for(pk.params, 0)
    eng.ensure_key(pk)
    rng = np.random.default_rng(28)
    # external product: tolerance 2^9 LSB vs the exact integer product
    idx = np.array([0, 3, 400, 819], np.int32)
    t = rng.integers(0, 2**32, (4, 2, N), dtype=np.uint64).astype(np.uint32)
    got = eng.batch_external_product(t, idx)
    for i, x, gg in zip(idx, t, got):
        exact = O.external_product_exact(ck.bootstrapping_key_time[i], x, P.l, P.bgbit, ck.decomposition_offset)
        assert signed_diff(gg, exact) <= 512
    # integer stage stays bit-exact with base = 32
    lv1 = rng.integers(0, 2**32, (5, N + 1), dtype=np.uint64).astype(np.uint32)
    assert np.array_equal(eng.batch_identity_key_switch(lv1), np.stack([O.identity_key_switching(ck, x) for x in lv1]))
    # decrypted messages identical to the CPU reference path
    msgs = np.concatenate([np.arange(16), rng.integers(0, 16, 16)])
    cts = sk.encrypt_lwe_message(msgs, 16, 60)
    for f in (lambda x: x % 16, lambda x: (x * x) % 16):
        lut = R.lut.Generator(16).generate_lookup_table(f)
        out = R.LutBootstrap().bootstrap_lut(cts, lut, pk)
        cpu = O.batch_bootstrap(ck, cts, testvec=lut.poly)
        want = np.array([f(int(m)) for m in msgs])
        assert np.array_equal(sk.decrypt_lwe_message(out, 16), want)
        assert np.array_equal(sk.decrypt_lwe_message(cpu, 16), want)
        # phases agree to 1/8 of a message step (2^32/32 = 2^27): at bgbit = 22 one f64 LSB flips a
        # digit, after which the two runs are two different valid noise realisations
        assert signed_diff(sk.phase(out), sk.phase(cpu)) < (1 << 24)


# SECURITY_UINT5 is run at m = 16: with N fixed at 1024 (params.rs:264-289 keeps the 1024-coefficient ring) the
# rounding of 1071 mask words to 2N positions alone mis-decodes ~5 % of m = 32 messages, on the CPU path too.
@pytest.mark.parametrize("setname,m", [("SECURITY_UINT2", 4), ("SECURITY_UINT3", 8), ("SECURITY_UINT5", 16)])
def test_pbs_other_uint_sets(O, setname, m):
    """The reference's other message-space sets (src/params.rs:177-289): wider key-switch bases (16, 64)
    and n up to 1071, through LutBootstrap with the message modulus of the set; decrypted messages equal
    f(x) and equal the CPU path's, phases agree to well under a message step."""
    import rs_tfhe_amd as R

    op = getattr(O, setname)
    sk, ck = oracle_keys(O, op)
    pk = _cloud_key(ck)
    eng = R.bootstrap.engine_for(pk.params, 0)
    eng.ensure_key(pk)
    rng = np.random.default_rng(33)
    msgs = np.concatenate([np.arange(m), rng.integers(0, m, 8)])
    cts = sk.encrypt_lwe_message(msgs, m, 61)
    for f in (lambda x: x % m, lambda x: (3 * x + 1) % m):
        lut = R.lut.Generator(m).generate_lookup_table(f)
        out = R.LutBootstrap().bootstrap_lut(cts, lut, pk)
        cpu = O.batch_bootstrap(ck, cts, testvec=lut.poly)
        want = np.array([f(int(x)) for x in msgs])
        assert np.array_equal(sk.decrypt_lwe_message(cpu, m), want)
        assert np.array_equal(sk.decrypt_lwe_message(out, m), want)
        # not bit-comparable (bgbit >= 18: one f64 LSB flips a digit and the two runs become two
        # different valid noise realisations); they must agree to 1/8 of a message step
        assert signed_diff(sk.phase(out), sk.phase(cpu)) < (1 << 32) // (2 * m) // 8


def test_xor_and_mux_80bit(O, keys80):
    """BASELINE config 5 parameter set (n = 550, t = 7)."""
    import rs_tfhe_amd as R

    sk, ck = keys80
    pk = _cloud_key(ck)
    eng = R.bootstrap.engine_for(pk.params, 0)
    eng.ensure_key(pk)
    A = np.array([0, 0, 1, 1, 1], bool)
    B = np.array([0, 1, 0, 1, 0], bool)
    Cc = np.array([1, 0, 1, 0, 0], bool)
    ca, cb, cc = sk.encrypt_bool(A, 1), sk.encrypt_bool(B, 2), sk.encrypt_bool(Cc, 3)
    x = eng.batch_gate(O.GATE_XOR, ca, cb)
    assert np.array_equal(x, O.batch_gate(ck, O.GATE_XOR, ca, cb))
    assert np.array_equal(sk.decrypt_bool(x), A ^ B)
    m = eng.batch_mux(ca, cb, cc, naive=True)
    assert np.array_equal(m, O.batch_mux(ck, ca, cb, cc, naive=True))
    assert np.array_equal(sk.decrypt_bool(m), np.where(A, B, Cc))
    assert np.array_equal(eng.batch_mux(ca, cb, cc, naive=False), O.batch_mux(ck, ca, cb, cc, naive=False))


def test_other_parameter_sets_bit_exact(O):
    """SECURITY_UINT1 (l=2, bgbit=10: the general rounding path, L=2 kernels) and SECURITY_110_BIT
    (t=8): whole bootstraps bit-identical to the CPU path, gates decrypt correctly."""
    import rs_tfhe_amd as R

    for op_params, seed in ((O.SECURITY_UINT1, 41), (O.SECURITY_110_BIT, 42)):
        sk, ck = oracle_keys(O, op_params, seed=seed)
        pk = _cloud_key(ck)
        eng = R.bootstrap.engine_for(pk.params, 0)
        eng.ensure_key(pk)
        A = np.array([1, 1, 0, 0, 1, 0, 1], bool)
        B = np.array([1, 0, 1, 0, 0, 0, 1], bool)
        ca, cb = sk.encrypt_bool(A, seed + 100), sk.encrypt_bool(B, seed + 200)
        for op in (O.GATE_NAND, O.GATE_XOR, O.GATE_ORYN):
            got = eng.batch_gate(op, ca, cb)
            assert np.array_equal(got, O.batch_gate(ck, op, ca, cb)), (op_params.name, op)
            exp = np.array([O.GATE_TRUTH[op](bool(a), bool(b)) for a, b in zip(A, B)])
            assert np.array_equal(sk.decrypt_bool(got), exp)
        tr = eng.batch_blind_rotate(ca[:3])
        assert np.array_equal(tr, O.batch_blind_rotate(ck, ca[:3]))


def test_extreme_parameter_shapes(O):
    """Boundary shapes of the run-time parametric engine: the largest supported LWE dimension
    (n = 1279: 5-wave key-switch blocks, 20.9 KB of LDS per blind-rotate wave) and the smallest
    (n = 1), with l = 2 and an odd key-switch depth; bit-exact vs the CPU path."""
    import rs_tfhe_amd as R
    from rs_tfhe_amd.params import SecurityParams

    for (n, l, bgbit, basebit, t, seed) in ((1279, 2, 8, 2, 5, 51), (1, 3, 6, 2, 3, 52), (64, 1, 16, 3, 4, 53)):
        # a 2^15-wide digit needs the (near) noise-free key of the UINT sets (params.rs:235-260), or the
        # phase noise itself wraps the torus and nothing is comparable
        op = O.Params(f"EDGE_{n}", n, l, bgbit, basebit, t, 2.0e-5, 2.0e-8 if bgbit <= 10 else 2.2e-16)
        sk, ck = O.keygen(op, seed)
        pp = SecurityParams(op.name, 0, n, l, bgbit, basebit, t, op.alpha_lv0, op.alpha_lv1)
        pk = R.CloudKey(pp, ck.bootstrapping_key, ck.key_switching_key, ck.decomposition_offset, ck.blind_rotate_testvec)
        eng = R.Engine(pp, 0)
        eng.load_cloud_key(pk)
        rng = np.random.default_rng(seed)
        a = rng.integers(0, 2**32, (3, n + 1), dtype=np.uint64).astype(np.uint32)
        b = rng.integers(0, 2**32, (3, n + 1), dtype=np.uint64).astype(np.uint32)
        got = eng.batch_gate(O.GATE_XOR, a, b)
        exp = O.batch_gate(ck, O.GATE_XOR, a, b)
        if bgbit + np.log2(2 * l) < 12:  # exact-product regime: 2l*N*(Bg/2)*2^31 < 2^52
            assert np.array_equal(got, exp), (n, l, bgbit)
        else:
            # inexact f64 products: a one-LSB difference can flip a decomposition digit, which swaps
            # in a different (uniformly random) key row -- ciphertext words then differ wholesale while
            # the PHASE moves by a noise term only.  Compare phases, as for SECURITY_UINT4.
            g_t, e_t = eng.batch_blind_rotate(a), O.batch_blind_rotate(ck, a)
            for x, y in zip(g_t, e_t):
                assert signed_diff(sk.trlwe_phase(x), sk.trlwe_phase(y)) < (1 << 24)
        lv1 = rng.integers(0, 2**32, (5, N + 1), dtype=np.uint64).astype(np.uint32)
        assert np.array_equal(eng.batch_identity_key_switch(lv1), np.stack([O.identity_key_switching(ck, x) for x in lv1]))
        eng.close()
    with pytest.raises(R._capi.TfheHipError):
        R.Engine(SecurityParams("TOO_BIG", 0, 1280, 3, 6, 2, 9, 1e-5, 1e-8), 0)


L1_EXACT_SHAPES = (  # (n, l, bgbit, basebit, t, seed)
    (64, 1, 10, 5, 3, 54),    # general rounding (1 + 10 + 9 + 31 = 51 is not < 51) in the exact regime (|x| < 2^51)
    (820, 1, 10, 5, 3, 55),   # SECURITY_UINT4's n, base-32 key switch
    (300, 1, 9, 2, 8, 56),    # l = 1 on the FAST rounding path (50 < 51): k_blind_rotate<1, true>
)


def _edge_keys(O, n, l, bgbit, basebit, t, seed):
    import rs_tfhe_amd as R
    from rs_tfhe_amd.params import SecurityParams

    op = O.Params(f"EDGE_{n}_{l}_{bgbit}", n, l, bgbit, basebit, t, 2.0e-5, 2.0e-8)
    sk, ck = oracle_keys(O, op, seed=seed, with_time=True)
    pp = SecurityParams(op.name, 0, n, l, bgbit, basebit, t, op.alpha_lv0, op.alpha_lv1)
    pk = R.CloudKey(pp, ck.bootstrapping_key, ck.key_switching_key, ck.decomposition_offset, ck.blind_rotate_testvec)
    return sk, ck, pp, pk


@pytest.mark.parametrize("shape", L1_EXACT_SHAPES, ids=lambda s: f"n{s[0]}_l{s[1]}_bg{s[2]}")
def test_l1_general_rounding_exact_regime_bit_exact(O, shape, monkeypatch):
    """l = 1 where the f64 product is EXACT (bgbit = 10: 2 * 1024 * 512 * 2^31 = 2^51 < 2^52), so the reference's
    `round() as i64 as u32` (klemsa.rs:145-146) has one right answer per word and any rounding slip shows: the
    general-rounding instantiations k_blind_rotate<1, false>, k_blind_rotate_wide2<1, false>, k_blind_rotate_pair<1, false>
    and k_external_product<1, false> -- which every reference parameter set with l = 1 runs, but only in the inexact
    regime (bgbit >= 15), where tests can compare by tolerance alone -- word for word against the CPU path AND against
    the exact integer product; (300, 1, 9) does the same for the FAST l = 1 instantiations."""
    import rs_tfhe_amd as R

    n, l, bgbit, basebit, t, seed = shape
    sk, ck, pp, pk = _edge_keys(O, *shape)
    rng = np.random.default_rng(seed)
    cts = rng.integers(0, 2**32, (21, n + 1), dtype=np.uint64).astype(np.uint32)
    cts2 = rng.integers(0, 2**32, (21, n + 1), dtype=np.uint64).astype(np.uint32)
    want_rot = O.batch_blind_rotate(ck, cts)
    want_boot = O.batch_bootstrap(ck, cts)
    want_gate = O.batch_gate(ck, O.GATE_XOR, cts, cts2)
    # the CPU path itself is exact here: its f64 blind rotation equals the exact-integer one
    for i in (0, 20):
        assert np.array_equal(want_rot[i], O.blind_rotate(ck, cts[i], exact=True))
    trl = rng.integers(0, 2**32, (6, 2, N), dtype=np.uint64).astype(np.uint32)
    idx = np.array([0, 1, n // 2, n - 1, 3 % n, 7 % n], np.int32)
    want_ep = np.stack([O.external_product_exact(ck.bootstrapping_key_time[i], x, l, bgbit, ck.decomposition_offset) for i, x in zip(idx, trl)])
    for name in BR_KERNEL_ENVS:
        _with_br_kernel(monkeypatch, name)
        eng = R.Engine(pp, 0)
        eng.load_cloud_key(pk)
        assert eng.rounding_mode == ("general" if bgbit == 10 else "fast")
        assert f"blind_rotate={name}[0,21)" in eng.describe_dispatch(21)
        assert np.array_equal(eng.batch_blind_rotate(cts), want_rot), (shape, name)
        assert np.array_equal(eng.batch_bootstrap(cts), want_boot), (shape, name)
        assert np.array_equal(eng.batch_gate(O.GATE_XOR, cts, cts2), want_gate), (shape, name)
        assert np.array_equal(eng.batch_external_product(trl, idx), want_ep), (shape, name)  # 0 LSB vs the exact product
        eng.close()


def test_rounding_mutation_is_caught():
    """The suite must NOTICE a one-LSB slip in the general rounding: the same l = 1 exact-regime test, run as a child
    process against a mutation build of the library (csrc/experiment.hpp TFHE_ABL_ROUND_LSB: round_product<false> returns
    one LSB too much on ~1/1024 of the words; `make -C rs-tfhe_amd/csrc mutation`, built by __graft_entry__.build()), has to
    FAIL on the general-rounding shapes -- with assertion errors of that test, not for any other reason -- and still pass
    on the FAST-path shape, which the mutation does not touch."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "rs-tfhe_amd", "libtfhe_v_round_lsb.so")
    # the mutation library is built here if it is missing or older than its sources (`make` decides): the product's
    # build does not depend on a library that is wrong by construction (__graft_entry__.build() makes it, non-fatally)
    subprocess.check_call(["make", "-C", os.path.join(root, "rs-tfhe_amd", "csrc"), "mutation"], stdout=subprocess.DEVNULL)
    assert os.path.exists(lib), "mutation build missing: make -C rs-tfhe_amd/csrc mutation"
    env = dict(os.environ, TFHE_HIP_LIB=lib, TFHE_HIP_ALLOW_EXPERIMENT="1")
    xml = os.path.join(root, "gpurun_out", "mutation_junit.xml")
    os.makedirs(os.path.dirname(xml), exist_ok=True)
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_parity.py"), "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "test_l1_general_rounding_exact_regime_bit_exact", "--junitxml", xml], cwd=root, env=env,
                       capture_output=True, text=True, timeout=1200)
    out = p.stdout + p.stderr
    assert p.returncode == 1, out[-3000:]
    # per-test outcomes from the junit file (not from pytest's summary line or the repr of its assertion text)
    import xml.etree.ElementTree as ET

    outcome, message = {}, {}
    for case in ET.parse(xml).getroot().iter("testcase"):
        fail = case.find("failure")
        err = case.find("error")
        outcome[case.get("name")] = "error" if err is not None else ("failed" if fail is not None else "passed")
        message[case.get("name")] = (fail.get("message") or "") + (fail.text or "") if fail is not None else ""
    general = [k for k in outcome if "n64_l1_bg10" in k or "n820_l1_bg10" in k]
    fast = [k for k in outcome if "n300_l1_bg9" in k]
    assert len(general) == 2 and len(fast) == 1, outcome
    for k in general:  # the general-rounding shapes FAIL, and they fail in the word-for-word comparison
        assert outcome[k] == "failed" and "AssertionError" in message[k] and "array_equal" in message[k], (k, outcome[k], message[k][-1500:])
    assert outcome[fast[0]] == "passed", (outcome, out[-3000:])  # the FAST-path shape is not touched by the mutation
    assert "ImportError" not in out and "Error loading" not in out


INEXACT_SETS = ["SECURITY_UINT2", "SECURITY_UINT3", "SECURITY_UINT4", "SECURITY_UINT5", "SECURITY_UINT6", "SECURITY_UINT7", "SECURITY_UINT8"]


@pytest.mark.parametrize("setname", INEXACT_SETS)
def test_external_product_error_relative_to_the_cpu_path(O, setname):
    """Where the f64 product is NOT exact (bgbit >= 15) the reference's own result carries an f64 error (~2^7
    LSB at bgbit = 22).  Instead of a fixed bound: on the same inputs, the GPU's error against the exact integer product
    must be no more than twice the CPU path's (klemsa.rs:119-150 restated in the oracle) -- per external product, 24
    products per set, every reference set with an inexact product."""
    import rs_tfhe_amd as R

    op = getattr(O, setname)
    sk, ck = oracle_keys(O, op, seed=88, with_time=True)
    pk = _cloud_key(ck)
    eng = R.bootstrap.engine_for(pk.params, 0)
    eng.ensure_key(pk)
    assert eng.rounding_mode == "general"
    rng = np.random.default_rng(88)
    count = 24
    idx = rng.integers(0, op.n, count).astype(np.int32)
    trl = rng.integers(0, 2**32, (count, 2, N), dtype=np.uint64).astype(np.uint32)
    got = eng.batch_external_product(trl, idx)
    worst = 0.0
    for i, x, g in zip(idx, trl, got):
        exact = O.external_product_exact(ck.bootstrapping_key_time[i], x, op.l, op.bgbit, ck.decomposition_offset)
        cpu = O.external_product_fft(ck.bootstrapping_key[i], x, op.l, op.bgbit, ck.decomposition_offset)
        e_gpu, e_cpu = signed_diff(g, exact), signed_diff(cpu, exact)
        assert e_gpu <= 2 * max(e_cpu, 1), (setname, int(i), e_gpu, e_cpu)
        worst = max(worst, e_gpu / max(e_cpu, 1))
    assert worst <= 2.0


def test_full_size_pbs_uint4(O, keys_uint4):
    """BASELINE configs[3] at full size: 65,536 DISTINCT ciphertexts through LutBootstrap::bootstrap_lut (m = 16,
    f = x^2 mod 16), SECURITY_UINT4, device-resident.  Every output decrypts to f(message); the same input at another
    batch position gives the same bits (no cross-ciphertext leakage) and a second run gives the same batch; 2,048
    outputs spread over the whole batch decrypt identically on the CPU path with phases within 1/8 of a message step."""
    import torch

    import rs_tfhe_amd as R

    sk, ck = keys_uint4
    pk = _cloud_key(ck)
    eng = R.bootstrap.engine_for(pk.params, 0)
    eng.ensure_key(pk)
    B = 65536
    rng = np.random.default_rng(44)
    msgs = rng.integers(0, 16, B)
    cts = sk.encrypt_lwe_message(msgs, 16, 4401)
    dup = np.array([0, 1, 777, 40000])  # the same ciphertexts again at the end of the batch
    cts[B - len(dup):] = cts[dup]
    msgs[B - len(dup):] = msgs[dup]
    lut = R.lut.Generator(16).generate_lookup_table(lambda x: (x * x) % 16)
    dev = torch.device("cuda:0")
    tin = torch.from_numpy(cts.view(np.int32)).to(dev)
    tlut = torch.from_numpy(lut.poly.view(np.int32)).to(dev)
    tout, tout2 = torch.empty_like(tin), torch.empty_like(tin)
    eng.batch_bootstrap_dev(tin, tout, testvec=tlut)
    eng.batch_bootstrap_dev(tin, tout2, testvec=tlut)
    torch.cuda.synchronize()
    assert torch.equal(tout, tout2)
    out = tout.cpu().numpy().view(np.uint32)
    n = pk.params.n
    assert np.array_equal(out[B - len(dup):], out[dup])
    # vectorised decrypt_lwe_message (tlwe.rs:111-126) of the whole batch
    phase = out[:, n] - (out[:, :n] * sk.key_lv0[None, :]).sum(axis=1, dtype=np.uint32)
    dec = ((phase.astype(np.float64) / 2.0**32) * 32.0 + 0.5).astype(np.int64) % 16
    assert np.array_equal(dec, (msgs * msgs) % 16)
    # 2,048 outputs spread over the whole batch vs the CPU path: identical messages, phases within 2^24
    # (1/8 of a message step; the f64 products differ by ~2^7 LSB at bgbit = 22)
    idx = np.unique(np.concatenate([np.arange(64), np.linspace(0, B - 1, 1920).astype(np.int64), np.arange(B - 64, B)]))
    assert len(idx) >= 2000
    cpu = O.batch_bootstrap(ck, cts[idx], testvec=lut.poly)
    assert np.array_equal(sk.decrypt_lwe_message(cpu, 16), sk.decrypt_lwe_message(out[idx], 16))
    assert np.abs((sk.phase(cpu) - sk.phase(out[idx])).view(np.int32)).max() < (1 << 24)


def test_full_size_mixed_circuit_80bit(O, keys80):
    """BASELINE configs[4] shape on one GPU: a mixed mux + xor circuit at SECURITY_80_BIT over a
    65,536-wide batch (65,536 mux_naive = 196,608 gates + 65,536 xor), two launches in all."""
    import torch

    import rs_tfhe_amd as R

    sk, ck = keys80
    pk = _cloud_key(ck)
    eng = R.bootstrap.engine_for(pk.params, 0)
    eng.ensure_key(pk)
    B, base = 65536, 4096
    rng = np.random.default_rng(45)
    bits = rng.integers(0, 2, (3, base)).astype(bool)
    ins0 = np.stack([sk.encrypt_bool(bits[w], 4500 + w) for w in range(3)])  # [3][base][n+1]
    ins = np.tile(ins0, (1, B // base, 1))
    c = R.Circuit(3)
    w_mux = c.mux_naive(0, 1, 2)
    w_xor = c.xor(0, 1)
    assert [len(l) for l in c.levels()] == [3, 1]
    dev = torch.device("cuda:0")
    wires = c.run_dev(eng, torch.from_numpy(ins.view(np.int32)).to(dev))
    torch.cuda.synchronize()
    n = pk.params.n
    for w, want in ((w_mux, np.where(bits[0], bits[1], bits[2])), (w_xor, bits[0] ^ bits[1])):
        out = wires[w].cpu().numpy().view(np.uint32)
        phase = out[:, n] - (out[:, :n] * sk.key_lv0[None, :]).sum(axis=1, dtype=np.uint32)
        assert np.array_equal(phase.view(np.int32) >= 0, np.tile(want, B // base))
        assert np.array_equal(out[:base], out[B - base:])
    ref = c.run_reference(lambda op, a, b: O.batch_gate(ck, op, a, b), ins0[:, :4])
    assert np.array_equal(wires[w_mux][:4].cpu().numpy().view(np.uint32), ref[w_mux])


def test_configs4_share_mux_and_xor_80bit(O, keys80):
    """BASELINE configs[4] exactly, per-GPU share of the 1M-gate batch: 131,072 gates at SECURITY_80_BIT, half
    `Gates::mux` in the reference's own formula (gates.rs:157-183: two bootstrap_without_key_switch, add, one full
    bootstrap -- quirk Q5: not a decryptable construction, reproduced bit for bit) and half hom_xor, all 327,680
    input ciphertexts distinct, as ONE circuit level: two blind-rotation launches and one key switch
    (circuit.mux_and_gates_dev).  Bar: 640 mux and 1,024 xor outputs spread over each half equal the CPU path word for
    word, the whole mux half equals the gate-by-gate device path (batch_mux_dev) word for word, the xor half decrypts."""
    import torch

    import rs_tfhe_amd as R

    sk, ck = keys80
    pk = _cloud_key(ck)
    eng = R.bootstrap.engine_for(pk.params, 0)
    eng.ensure_key(pk)
    M = X = 65536
    rng = np.random.default_rng(48)
    bits = rng.integers(0, 2, (5, M)).astype(bool)
    a, b, c, xa, xb = (sk.encrypt_bool(bits[w], 4800 + w) for w in range(5))
    dev = torch.device("cuda:0")
    ta, tb, tc, txa, txb = (torch.from_numpy(x.view(np.int32)).to(dev) for x in (a, b, c, xa, xb))
    codes = torch.full((X,), R.engine.XOR, dtype=torch.uint8, device=dev)
    eng.kernel_times()
    eng.set_profiling(True)
    mo, xo = R.circuit.mux_and_gates_dev(eng, ta, tb, tc, codes, txa, txb)
    torch.cuda.synchronize()
    eng.set_profiling(False)
    kt = eng.kernel_times()
    # two blind-rotation launches (more only if small calls arrived on this context in the last 250 ms: bulk launches then
    # go out in chunks, include/tfhe_hip.h) and one key switch
    assert kt["blind_rotate_launches"] >= 2 and kt["key_switch_launches"] == 1 and kt["bootstraps"] == 2 * M + M + X
    mux, xor = mo.cpu().numpy().view(np.uint32), xo.cpu().numpy().view(np.uint32)
    # 640 mux outputs (3 bootstraps each on the CPU) and 1,024 xor outputs spread over each half vs the CPU path
    sl = np.unique(np.r_[0:24, np.linspace(0, M - 1, 592).astype(np.int64), M - 24:M])
    assert np.array_equal(mux[sl], O.batch_mux(ck, a[sl], b[sl], c[sl], naive=False))
    sx = np.unique(np.r_[0:24, np.linspace(0, X - 1, 976).astype(np.int64), X - 24:X])
    assert np.array_equal(xor[sx], O.batch_gate(ck, O.GATE_XOR, xa[sx], xb[sx]))
    assert np.array_equal(sk.decrypt_bool(xor), bits[3] ^ bits[4])
    ref = torch.empty_like(ta)
    eng.batch_mux_dev(ta, tb, tc, ref, naive=False)
    torch.cuda.synchronize()
    assert torch.equal(ref, mo)
    # the same level through the host-array entry points on a ragged slice (tfhe_hip_batch_gates_mixed_nks)
    k = 37
    g1 = np.r_[np.full(k, R.engine.AND), np.full(k, R.engine.ANDNY)].astype(np.uint8)
    u = eng.batch_gates_mixed(g1, np.concatenate([a[:k], a[:k]]), np.concatenate([b[:k], c[:k]]), keyswitch=False)
    assert np.array_equal(eng.batch_gate(R.engine.OR, u[:k], u[k:]), mux[:k])
    with pytest.raises(R._capi.TfheHipError):
        eng.batch_gates_mixed(np.full(2 * k, 11, np.uint8), np.concatenate([a[:k], a[:k]]), np.concatenate([b[:k], c[:k]]), keyswitch=False)


def test_soak_gates_vs_cpu_path_128bit(O, eng128, keys128):
    """Soak: every gate x many ciphertexts, a large batch through the LDS-ring key switch,
    and uniformly random words instead of encryptions, against the CPU path under the same key: not one
    ciphertext may differ in any word, every gate output decrypts to its truth table.  ~18 k bootstraps by
    default (about half a minute of CPU checking on the box's cores); TFHE_SOAK_FULL=1 runs the 70,440 of the
    round-1 one-off."""
    import os

    sk, ck = keys128
    full = os.environ.get("TFHE_SOAK_FULL") == "1"
    per_gate, big, rand = (4096, 16384, 4096) if full else (1024, 6144, 1024)
    rng = np.random.default_rng(4700)
    bad = 0
    for op in range(10):
        A, B = rng.integers(0, 2, per_gate).astype(bool), rng.integers(0, 2, per_gate).astype(bool)
        ca, cb = sk.encrypt_bool(A, 47000 + 2 * op), sk.encrypt_bool(B, 47001 + 2 * op)
        got = eng128.batch_gate(op, ca, cb)
        bad += int((got != O.batch_gate(ck, op, ca, cb)).any(axis=1).sum())
        want = np.array([O.GATE_TRUTH[op](bool(x), bool(y)) for x, y in zip(A, B)])
        assert np.array_equal(sk.decrypt_bool(got), want), O.GATE_NAMES[op]
    A, B = rng.integers(0, 2, big).astype(bool), rng.integers(0, 2, big).astype(bool)
    ca, cb = sk.encrypt_bool(A, 47100), sk.encrypt_bool(B, 47101)
    got = eng128.batch_gate(O.GATE_NAND, ca, cb)
    bad += int((got != O.batch_gate(ck, O.GATE_NAND, ca, cb)).any(axis=1).sum())
    assert np.array_equal(sk.decrypt_bool(got), ~(A & B))
    ra = rng.integers(0, 2**32, (rand, 701), dtype=np.uint64).astype(np.uint32)  # not encryptions of anything
    rb = rng.integers(0, 2**32, (rand, 701), dtype=np.uint64).astype(np.uint32)
    rx = eng128.batch_gate(O.GATE_XOR, ra, rb)
    bad += int((rx != O.batch_gate(ck, O.GATE_XOR, ra, rb)).any(axis=1).sum())
    # the same random words in pieces small enough for the latency kernels (<= 3 x #CUs): same bits as the batch kernels
    lo = 0
    for piece in (1, 37, 250, 700):
        bad += int((eng128.batch_gate(O.GATE_XOR, ra[lo:lo + piece], rb[lo:lo + piece]) != rx[lo:lo + piece]).any(axis=1).sum())
        lo += piece
    assert bad == 0, f"{bad} ciphertexts differ from the CPU path"


def test_contexts_views_and_pools_give_their_memory_back(O, keys128):
    """Create / use / destroy cycles of contexts, key views and pools (host API, device API, every batch-size regime
    so that all scratch buffers get allocated): device memory in use returns to where it started."""
    import torch

    import rs_tfhe_amd as R

    sk, ck = keys128
    pk = _cloud_key(ck)
    A = np.resize(np.array([1, 0, 1, 1, 0], bool), 1100)
    ca, cb = sk.encrypt_bool(A, 7301), sk.encrypt_bool(~A, 7302)

    def cycle():
        eng = R.Engine(pk.params, 0)
        eng.load_cloud_key(pk)
        view = eng.new_key_view()
        view.load_cloud_key(pk)
        for count in (1, 300, 1100):  # one and two ciphertexts per workgroup, batch kernel + matrix-core key switch
            eng.batch_gate(O.GATE_NAND, ca[:count], cb[:count])
        view.batch_gate(O.GATE_XOR, ca[:5], cb[:5])
        view.close()
        eng.close()
        pool = R.Pool(pk.params, [0, 0])
        pool.load_cloud_key(pk)
        pv = pool.new_key_view()
        pv.load_cloud_key(pk)
        pv.batch_gate(O.GATE_AND, ca[:600], cb[:600])
        pv.close()
        pool.close()

    cycle()  # first use pays for lazily created runtime state (streams, module load)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(0)
    for _ in range(3):
        cycle()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info(0)
    assert free0 - free1 < 8 << 20, f"{(free0 - free1) >> 20} MiB not returned after three create/destroy cycles"


def test_bad_gate_code_on_the_device_is_reported(O, eng128, keys128):
    """tfhe_hip_batch_gates_mixed_dev cannot read the codes on the host: a code outside tfhe_hip_gate is treated as
    COPY by the kernel AND raises a device-side flag that the next tfhe_hip_synchronize returns as EINVAL (once).
    The host-array entry point rejects the same batch up front."""
    import torch

    import rs_tfhe_amd as R

    sk, ck = keys128
    A = np.array([1, 0, 1, 1, 0], bool)
    ca, cb = sk.encrypt_bool(A, 4900), sk.encrypt_bool(~A, 4901)
    dev = torch.device("cuda:0")
    ta, tb = torch.from_numpy(ca.view(np.int32)).to(dev), torch.from_numpy(cb.view(np.int32)).to(dev)
    to = torch.empty_like(ta)
    eng128.synchronize()
    codes = torch.tensor([O.GATE_NAND, 200, O.GATE_XOR, 11, O.GATE_COPY], dtype=torch.uint8, device=dev)
    eng128.batch_gates_mixed_dev(codes, ta, tb, to)
    with pytest.raises(R._capi.TfheHipError, match="gate code"):
        eng128.synchronize()
    eng128.synchronize()  # reported once
    out = to.cpu().numpy().view(np.uint32)
    good = eng128.batch_gates_mixed(np.array([O.GATE_NAND, O.GATE_COPY, O.GATE_XOR, O.GATE_COPY, O.GATE_COPY], np.uint8), ca, cb)
    assert np.array_equal(out, good)
    with pytest.raises(R._capi.TfheHipError):
        eng128.batch_gates_mixed(codes.cpu().numpy(), ca, cb)
    eng128.batch_gates_mixed_dev(torch.zeros(5, dtype=torch.uint8, device=dev), ta, tb, to)
    eng128.synchronize()  # a clean launch leaves no flag behind


def test_two_threads_two_keys_through_the_gates_api():
    """`Gates` / `gates.batch_*` take `cloud_key` per call like the reference (`&CloudKey`, Send + Sync strategies,
    bootstrap/mod.rs:23): two threads alternating between two keys must each compute under their own key.  Each
    key is a key view (`tfhe_hip_key_create`) of the ONE context of its parameter set and device
    (bootstrap.keyed_engine): the call names its key by the handle it is made on.  A shared context whose key is
    "ensured" and then used in two steps fails this."""
    import threading

    import rs_tfhe_amd as R

    P = R.params.SECURITY_128_BIT
    sks = [R.SecretKey.new(P, seed=5100 + i) for i in range(2)]
    cks = [R.CloudKey.new(sk, seed=5200 + i) for i, sk in enumerate(sks)]
    bad, errs = [0, 0], []

    def worker(t):
        try:
            sk, ck = sks[t], cks[t]
            rng = np.random.default_rng(5300 + t)
            g = R.Gates()
            for it in range(5):
                A, B = rng.integers(0, 2, 6).astype(bool), rng.integers(0, 2, 6).astype(bool)
                ca, cb = sk.encrypt_bool(A, seed=int(rng.integers(1 << 30))), sk.encrypt_bool(B, seed=int(rng.integers(1 << 30)))
                bad[t] += int((sk.decrypt_bool(R.gates.batch_nand(ca, cb, ck)) != ~(A & B)).sum())
                bad[t] += int(bool(sk.decrypt_bool(g.xor(ca[0], cb[0], ck))[0]) != bool(A[0] ^ B[0]))
                bad[t] += int((sk.decrypt_bool(g.mux_naive(ca, cb, ca, ck)) != np.where(A, B, A)).sum())
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errs, errs
    assert bad == [0, 0], bad
    # ONE context for the parameter set on this device; both keys stayed resident as key views of it
    base = R.bootstrap._engines[(P, 0)]
    views = R.bootstrap._views[(P, 0)]
    assert all(v._parent is base for v in views.values())
    assert sum(v._key is cks[0] for v in views.values()) == 1 and sum(v._key is cks[1] for v in views.values()) == 1


def test_key_views_several_keys_on_one_context(O, eng128, keys128):
    """tfhe_hip_key_create (include/tfhe_hip.h): a context holds several resident cloud keys; a call names its key by
    the handle it is made on (the reference's `&CloudKey` argument, bootstrap/mod.rs:23-38).  Two views + the
    context's own key, interleaved and from two threads: every result equals the CPU path under THAT key; load ->
    export through a view is the identity; a pool view does the same on every member."""
    import threading

    import rs_tfhe_amd as R

    sk1, ck1 = keys128
    sk2, ck2 = oracle_keys(O, O.SECURITY_128_BIT, seed=4321)
    pk2 = _cloud_key(ck2)
    v1, v2 = eng128.new_key_view(), eng128.new_key_view()
    assert v1._lib.tfhe_hip_key_parent(v1._ctx) == eng128._ctx.value and not v1._lib.tfhe_hip_key_is_loaded(v1._ctx)
    with pytest.raises(R._capi.TfheHipError):  # an empty view has no key
        v1.batch_gate(O.GATE_NAND, sk1.encrypt_bool([True], 1), sk1.encrypt_bool([True], 2))
    v1.load_cloud_key(_cloud_key(ck1))
    v2.load_cloud_key(pk2)
    A = np.array([1, 0, 1, 1, 0, 0, 1], bool)
    B = np.array([1, 1, 0, 1, 0, 1, 0], bool)
    in1 = (sk1.encrypt_bool(A, 7101), sk1.encrypt_bool(B, 7102))
    in2 = (sk2.encrypt_bool(A, 7103), sk2.encrypt_bool(B, 7104))
    want1 = O.batch_gate(ck1, O.GATE_XOR, *in1)
    want2 = O.batch_gate(ck2, O.GATE_XOR, *in2)
    for _ in range(2):  # interleaved: no call leaks its key into the next
        assert np.array_equal(v2.batch_gate(O.GATE_XOR, *in2), want2)
        assert np.array_equal(eng128.batch_gate(O.GATE_XOR, *in1), want1)  # the context's own key is untouched
        assert np.array_equal(v1.batch_gate(O.GATE_XOR, *in1), want1)
    assert np.array_equal(sk2.decrypt_bool(want2), A ^ B)
    errs = []

    def worker(view, ins, want):
        try:
            for _ in range(6):
                if not np.array_equal(view.batch_gate(O.GATE_XOR, *ins), want):
                    errs.append("wrong key")
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    ths = [threading.Thread(target=worker, args=a) for a in ((v1, in1, want1), (v2, in2, want2), (eng128, in1, want1))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    back = v2.export_cloud_key()
    assert np.array_equal(back.bootstrapping_key.reshape(-1), np.asarray(ck2.bootstrapping_key).reshape(-1))
    nz = np.asarray(ck2.key_switching_key).reshape(N, -1, 4, 701)[:, :, 1:, :]
    assert np.array_equal(back.key_switching_key.reshape(N, -1, 4, 701)[:, :, 1:, :], nz)
    v2.close()
    assert np.array_equal(v1.batch_gate(O.GATE_XOR, *in1), want1)  # closing one view leaves the others alone
    v1.close()
    # wrong destroy order (parent before its view): the context stays alive until its last view goes
    par = R.Engine(R.params.SECURITY_128_BIT, 0)
    orphan = par.new_key_view()
    orphan.load_cloud_key(pk2)
    par._lib.tfhe_hip_ctx_destroy(par._ctx)
    par._ctx, par._views = None, []
    assert np.array_equal(orphan.batch_gate(O.GATE_XOR, *in2), want2)
    orphan.close()
    # the same through a pool: two keys on the members of one pool (two contexts on this GPU)
    pool = R.Pool(R.params.SECURITY_128_BIT, [0, 0])
    pool.load_cloud_key(_cloud_key(ck1))
    pv = pool.new_key_view()
    pv.load_cloud_key(pk2)
    big = 600  # > 256: both members take a shard
    Ab, Bb = np.resize(A, big), np.resize(B, big)
    i1 = (sk1.encrypt_bool(Ab, 7201), sk1.encrypt_bool(Bb, 7202))
    i2 = (sk2.encrypt_bool(Ab, 7203), sk2.encrypt_bool(Bb, 7204))
    assert np.array_equal(sk2.decrypt_bool(pv.batch_gate(O.GATE_NAND, *i2)), ~(Ab & Bb))
    assert np.array_equal(sk1.decrypt_bool(pool.batch_gate(O.GATE_NAND, *i1)), ~(Ab & Bb))
    assert np.array_equal(pv.batch_gate(O.GATE_NAND, i2[0][:9], i2[1][:9]), O.batch_gate(ck2, O.GATE_NAND, i2[0][:9], i2[1][:9]))
    pv.close()
    pool.close()


def test_pool_key_replication_transports(O, keys128, monkeypatch):
    """The pool replicates its key by one grouped ncclBroadcast per key buffer when its devices are distinct (librccl
    opened at run time) and by hipMemcpyPeer otherwise.  One GPU here: a pool of ONE member forced through the RCCL
    path (TFHE_HIP_POOL_RCCL=2: communicator of one rank -- symbol loading, call sequence, stream handling) and a
    pool that repeats the device (peer copies); both must evaluate under the right key afterwards."""
    import rs_tfhe_amd as R

    sk, ck = keys128
    pk = _cloud_key(ck)
    A = np.array([1, 0, 1, 1, 0], bool)
    B = np.array([1, 1, 0, 1, 0], bool)
    ca, cb = sk.encrypt_bool(A, 7301), sk.encrypt_bool(B, 7302)
    want = O.batch_gate(ck, O.GATE_NAND, ca, cb)
    monkeypatch.setenv("TFHE_HIP_POOL_RCCL", "2")
    one = R.Pool(R.params.SECURITY_128_BIT, [0])
    one.load_cloud_key(pk)
    assert one.key_transport == "rccl"
    assert np.array_equal(one.batch_gate(O.GATE_NAND, ca, cb), want)
    one.close()
    two = R.Pool(R.params.SECURITY_128_BIT, [0, 0])
    two.load_cloud_key(pk)
    assert two.key_transport == "peer-copy"
    big = np.resize(np.arange(600) % 5, 600)
    assert np.array_equal(two.batch_gate(O.GATE_NAND, ca[big], cb[big]), want[big])
    two.close()


def test_pinned_host_buffers_run_in_place(O, eng128, keys128):
    """tfhe_hip_host_alloc: when every ciphertext operand of a host-API call is pinned memory the kernels read and
    write it in place (no staging); results must be the same words as through pageable arrays, for every entry
    point with the fast path, at a small (latency kernels) and a large (batch kernels) count, and through the pool."""
    import rs_tfhe_amd as R
    from rs_tfhe_amd.engine import pinned_copy, pinned_empty

    sk, ck = keys128
    rng = np.random.default_rng(5400)
    for count in (3, 1500):
        A, B, Cc = (rng.integers(0, 2, count).astype(bool) for _ in range(3))
        ca, cb, cc = sk.encrypt_bool(A, 5401), sk.encrypt_bool(B, 5402), sk.encrypt_bool(Cc, 5403)
        pa, pb, pc = pinned_copy(ca), pinned_copy(cb), pinned_copy(cc)
        po = pinned_empty(ca.shape)
        po[...] = 0xDEADBEEF
        want = eng128.batch_gate(O.GATE_NAND, ca, cb)
        got = eng128.batch_gate(O.GATE_NAND, pa, pb, out=po)
        assert got is po and np.array_equal(po, want) and np.array_equal(sk.decrypt_bool(po), ~(A & B))
        assert np.array_equal(pa, ca) and np.array_equal(pb, cb)  # inputs untouched
        # mixed operands (one pageable): falls back to staging, same words
        assert np.array_equal(eng128.batch_gate(O.GATE_NAND, pa, cb, out=pinned_empty(ca.shape)), want)
        lib, ctx = eng128._lib, eng128._ctx
        import ctypes as C

        def p(x):
            return x.ctypes.data_as(C.c_void_p)

        po[...] = 0
        codes = rng.integers(0, 11, count).astype(np.uint8)
        assert lib.tfhe_hip_batch_gates_mixed(ctx, p(codes), p(pa), p(pb), p(po), count) == 0
        assert np.array_equal(po, eng128.batch_gates_mixed(codes, ca, cb))
        po[...] = 0
        assert lib.tfhe_hip_batch_bootstrap(ctx, p(pa), None, 0, 1, p(po), count) == 0
        assert np.array_equal(po, eng128.batch_bootstrap(ca))
        tv = pinned_copy(rng.integers(0, 2**32, (count, 2, N), dtype=np.uint64).astype(np.uint32))
        assert lib.tfhe_hip_batch_bootstrap(ctx, p(pa), p(tv), 1, 0, p(po), count) == 0
        assert np.array_equal(po, eng128.batch_bootstrap(ca, np.array(tv), keyswitch=False))
        for naive in (1, 0):
            po[...] = 0
            assert lib.tfhe_hip_batch_mux(ctx, naive, p(pa), p(pb), p(pc), p(po), count) == 0
            assert np.array_equal(po, eng128.batch_mux(ca, cb, cc, naive=bool(naive)))
    pool = R.Pool(eng128.params, [0, 0])
    pool.load_cloud_key(_cloud_key(ck))
    po[...] = 0
    assert np.array_equal(pool.batch_gate(O.GATE_NAND, pa, pb, out=po), want)  # shards are interior pointers of pinned arrays
    pool.close()


# the three blind-rotation kernels, each forced at every batch size (TFHE_HIP_BR_KERNEL, include/tfhe_hip.h): eight waves
# per ciphertext (default up to #CUs), two ciphertexts per eight-wave workgroup (default for #CUs < count <= 2 #CUs),
# the batch kernel
BR_KERNEL_ENVS = {
    "single": {"TFHE_HIP_BR_KERNEL": "single"},
    "pair": {"TFHE_HIP_BR_KERNEL": "pair"},
    "batch": {"TFHE_HIP_BR_KERNEL": "batch"},
}


def _with_br_kernel(monkeypatch, name):
    for k, v in BR_KERNEL_ENVS[name].items():
        monkeypatch.setenv(k, v)


def test_latency_and_batch_kernels_agree(O, keys128, monkeypatch):
    """Small batches go through the latency kernels -- eight waves per ciphertext (blind_rotate_wide.hpp) or two
    ciphertexts per eight-wave workgroup (k_blind_rotate_pair, odd counts included) -- larger ones through the
    one-wave-per-ciphertext batch kernel: all three must give the oracle's bits, for every output form."""
    import rs_tfhe_amd as R

    sk, ck = keys128
    pk = _cloud_key(ck)
    rng = np.random.default_rng(46)
    A = rng.integers(0, 2, 41).astype(bool)
    B = rng.integers(0, 2, 41).astype(bool)
    ca, cb = sk.encrypt_bool(A, 4600), sk.encrypt_bool(B, 4601)
    outs = {}
    for name in BR_KERNEL_ENVS:
        _with_br_kernel(monkeypatch, name)
        eng = R.Engine(pk.params, 0)
        eng.load_cloud_key(pk)
        assert f"blind_rotate={name}[0,41)" in eng.describe_dispatch(41)
        outs[name] = (eng.batch_gate(O.GATE_NAND, ca, cb), eng.batch_blind_rotate(ca[:5]),
                      eng.batch_bootstrap(ca[:5], keyswitch=False),
                      eng.batch_gates_mixed(np.arange(41, dtype=np.uint8) % 10, ca, cb),
                      eng.batch_gate(O.GATE_XOR, ca[:1], cb[:1]))
        eng.close()
    for other in ("pair", "batch"):
        for x, y in zip(outs["single"], outs[other]):
            assert np.array_equal(x, y), other
    assert np.array_equal(outs["single"][0], O.batch_gate(ck, O.GATE_NAND, ca, cb))
    assert np.array_equal(outs["single"][1], O.batch_blind_rotate(ck, ca[:5]))
    assert np.array_equal(outs["single"][2], O.batch_bootstrap(ck, ca[:5], keyswitch=False))


def test_pair_kernel_and_tail_launches_in_the_default_dispatch(O, keys128):
    """Default dispatch on an MI355X (N = 256 CUs): N < count <= 2N takes k_blind_rotate_pair (301, odd); 2N < count
    <= 3N the first 2N as pairs and the rest one per workgroup (700); a tail of up to 2N above a multiple of 4N runs
    as a latency-kernel launch beside the batch kernel's (1,100 = 1,024 + 76; 1,400 = 1,024 + 376 as pairs).  All
    against the CPU path."""
    import rs_tfhe_amd as R

    sk, ck = keys128
    pk = _cloud_key(ck)
    eng = R.bootstrap.engine_for(pk.params, 0)
    eng.ensure_key(pk)
    rng = np.random.default_rng(461)
    for count, op in ((301, O.GATE_NOR), (700, O.GATE_XOR), (1100, O.GATE_NAND), (1400, O.GATE_ORYN)):
        A, B = rng.integers(0, 2, count).astype(bool), rng.integers(0, 2, count).astype(bool)
        ca, cb = sk.encrypt_bool(A, 4610 + count), sk.encrypt_bool(B, 4611 + count)
        got = eng.batch_gate(op, ca, cb)
        assert np.array_equal(got, O.batch_gate(ck, op, ca, cb)), count
        want = np.array([O.GATE_TRUTH[op](bool(a), bool(b)) for a, b in zip(A, B)])
        assert np.array_equal(sk.decrypt_bool(got), want), count


@pytest.mark.gpu
@pytest.mark.parametrize("setname", ["SECURITY_UINT4", "SECURITY_UINT1", "SECURITY_UINT3"])
def test_latency_kernels_same_bits_as_batch_kernel_inexact_sets(O, setname, monkeypatch):
    """Where the f64 products are NOT exact (bgbit = 22 / 15) the result depends on the order of the floating-point
    operations: the eight-wave latency kernels (one and two ciphertexts per workgroup) keep the batch kernel's row order
    and FMA sequence, so a ciphertext's bits do not depend on the size of the batch it arrives in (l = 1, 2 and 1)."""
    import rs_tfhe_amd as R

    sk, ck = oracle_keys(O, getattr(O, setname), seed=77)
    pk = _cloud_key(ck)
    m = 4
    msgs = 1 + np.arange(12) % 3
    cts = sk.encrypt_lwe_message(msgs, m, seed=7700)
    outs = {}
    for name in BR_KERNEL_ENVS:
        _with_br_kernel(monkeypatch, name)
        eng = R.Engine(pk.params, 0)
        eng.load_cloud_key(pk)
        outs[name] = (eng.batch_blind_rotate(cts[:11]), eng.batch_bootstrap(cts[:11], keyswitch=True))
        eng.close()
    for other in ("pair", "batch"):
        for x, y in zip(outs["single"], outs[other]):
            assert np.array_equal(x, y), (setname, other)
    for k in outs:  # phases 1/8, 1/4, 3/8 are all in the positive half: the sign test vector gives +1/8 = true
        assert sk.decrypt_bool(outs[k][1]).all(), (setname, k)


# ---- mixed-gate batches and levelised circuits ---------------------------------------------
def test_mixed_gate_batch(O, eng128, keys128):
    """Per-ciphertext gate selectors: one launch, ten different gates."""
    sk, ck = keys128
    rng = np.random.default_rng(30)
    gates = np.array(list(range(10)) * 3 + [O.GATE_COPY, O.GATE_NAND], np.uint8)
    A = rng.integers(0, 2, len(gates)).astype(bool)
    B = rng.integers(0, 2, len(gates)).astype(bool)
    ca, cb = sk.encrypt_bool(A, 700), sk.encrypt_bool(B, 701)
    got = eng128.batch_gates_mixed(gates, ca, cb)
    for i, g in enumerate(gates):
        exp = O.batch_gate(ck, int(g), ca[i], cb[i] if g != O.GATE_COPY else None)[0]
        assert np.array_equal(got[i], exp), f"gate {g} at {i}"
    from rs_tfhe_amd import _capi

    with pytest.raises(_capi.TfheHipError):
        eng128.batch_gates_mixed(np.array([77], np.uint8), ca[:1], cb[:1])


def test_ripple_carry_adder_circuit(O, eng128, keys128):
    """examples/add_two_numbers.rs: 4-bit add of 6 independent input pairs, levelised on the GPU,
    vs the same DAG evaluated gate by gate on the CPU oracle (bit-exact) and vs plain integers."""
    import rs_tfhe_amd as R

    sk, ck = keys128
    nbits, B = 4, 6
    rng = np.random.default_rng(31)
    xs, ys = rng.integers(0, 16, B), rng.integers(0, 16, B)
    c = R.Circuit(2 * nbits + 1)
    a_w, b_w, cin = list(range(nbits)), list(range(nbits, 2 * nbits)), 2 * nbits
    sum_w, carry_w = c.add(a_w, b_w, cin)
    assert len(c.gates) == 5 * nbits and len(c.levels()) == 2 * nbits + 1  # carry chain: 2 levels per bit
    bits = np.zeros((2 * nbits + 1, B), bool)
    for i in range(nbits):
        bits[i] = (xs >> i) & 1
        bits[nbits + i] = (ys >> i) & 1
    inputs = np.stack([sk.encrypt_bool(bits[w], 800 + w) for w in range(2 * nbits + 1)])
    wires = c.run(eng128, inputs)
    ref = c.run_reference(lambda op, a, b: O.batch_gate(ck, op, a, b), inputs)
    assert np.array_equal(wires, ref)
    total = np.zeros(B, np.int64)
    for i, w in enumerate(sum_w):
        total += sk.decrypt_bool(wires[w]).astype(np.int64) << i
    total += sk.decrypt_bool(wires[carry_w]).astype(np.int64) << nbits
    assert np.array_equal(total, xs + ys)
    # mux_naive as a circuit == the batched mux_naive entry point
    m = R.Circuit(3)
    out_w = m.mux_naive(0, 1, 2)
    trip = np.stack([sk.encrypt_bool(rng.integers(0, 2, 5).astype(bool), 900 + k) for k in range(3)])
    assert np.array_equal(m.run(eng128, trip)[out_w], eng128.batch_mux(trip[0], trip[1], trip[2], naive=True))


# ---- cloud key: export round trip, generation on the GPU ---------------------------------------
def test_cloud_key_export_is_identity(O, eng128, keys128):
    """load -> export reproduces the uploaded key bit-for-bit (the engine-order permutation and
    the 2^-10 pre-scale are exact; k=0 key-switch rows are zero on both sides)."""
    sk, ck = keys128
    out = eng128.export_cloud_key()
    assert np.array_equal(out.bootstrapping_key, ck.bootstrapping_key)
    assert np.array_equal(out.key_switching_key, ck.key_switching_key)
    assert out.decomposition_offset == ck.decomposition_offset
    assert np.array_equal(out.blind_rotate_testvec, ck.blind_rotate_testvec)


def test_cloud_key_export_is_identity_general_rounding_sets(O, keys_uint4):
    """The same for a set that takes the general rounding (bgbit 22): its engine key carries a further 2^-32
    (fft512.hpp key_scale / round_product) -- a power of two, so load -> export is still the identity -- and a key
    generated on the GPU, exported and loaded into a second context gives the same ciphertext words."""
    import rs_tfhe_amd as R

    sk, ck = keys_uint4
    pk = _cloud_key(ck)
    eng = R.Engine(pk.params, 0)
    eng.load_cloud_key(pk)
    out = eng.export_cloud_key()
    assert np.array_equal(out.bootstrapping_key, ck.bootstrapping_key)
    assert np.array_equal(out.key_switching_key, ck.key_switching_key)
    gen = eng.new_key_view()
    gen.gen_cloud_key(sk.key_lv0, sk.key_lv1, 4321)
    gk = gen.export_cloud_key()
    other = eng.new_key_view()
    other.load_cloud_key(gk)
    assert np.array_equal(other.export_cloud_key().bootstrapping_key, gk.bootstrapping_key)
    msgs = np.arange(200) % 16
    cts = sk.encrypt_lwe_message(msgs, 16, 77)
    lut = R.lut.Generator(16).generate_lookup_table(lambda x: (3 * x + 1) % 16)
    a, b = gen.batch_bootstrap(cts, testvec=lut.poly), other.batch_bootstrap(cts, testvec=lut.poly)
    assert np.array_equal(a, b)
    assert np.array_equal(sk.decrypt_lwe_message(a, 16), (3 * msgs + 1) % 16)
    for e in (gen, other, eng):
        e.close()


def test_gpu_key_generation(O, keys128):
    """CloudKey::new on the GPU (key.rs:59-66).  RNG streams differ from the host generator, so the
    bar is structural + functional: every sampled BSK row is a TRLWE encryption of the right gadget
    value under s1, every sampled KSK row a TLWE encryption of k*s1[i]/base^(j+1) under s0, noise at
    the requested alpha; gates evaluated with the generated key decrypt correctly; the seed fixes the key."""
    import rs_tfhe_amd as R

    sk, _ = keys128
    P = R.params.SECURITY_128_BIT
    eng = R.Engine(P, 0)
    eng.gen_cloud_key(sk.key_lv0, sk.key_lv1, seed=2025)
    gk = eng.export_cloud_key()
    assert gk.decomposition_offset == 0x82080000
    assert not gk.blind_rotate_testvec[0].any() and (gk.blind_rotate_testvec[1] == 0x20000000).all()
    # BSK rows: spectrum -> torus polys (oracle inverse FFT) -> phase = b - a (*) s1
    for i in (0, 1, 350, 699):
        for r in range(2 * P.l):
            a = O.klemsa_fft(gk.bootstrapping_key[i, r, 0])
            b = O.klemsa_fft(gk.bootstrapping_key[i, r, 1])
            want_a = np.zeros(N, np.uint32)
            want_b = np.zeros(N, np.uint32)
            g = np.uint32(int(sk.key_lv0[i]) * (1 << (32 - P.bgbit * (r % P.l + 1))))
            # gadget sits on a[0] for r < l, on b[0] for r >= l (trgsw.rs:44-47); the TRLWE part encrypts 0:
            # b - a (*) s1 == noise (+ gadget if r >= l), with the a-gadget contributing -g*s1 to the phase
            phase = (b - O.negacyclic_schoolbook(a, sk.key_lv1)).astype(np.uint32)
            if r < P.l:
                phase = (phase + O.negacyclic_schoolbook(np.concatenate([[g], np.zeros(N - 1, np.uint32)]).astype(np.uint32), sk.key_lv1)).astype(np.uint32)
            else:
                phase[0] -= g
            noise = phase.view(np.int32).astype(np.float64) / 2.0**32
            assert np.abs(noise).max() < 8 * P.alpha_lv1 + 2.0**-31, (i, r, np.abs(noise).max())
            assert noise.std() > 0.3 * P.alpha_lv1  # it is noise, not zeros
    # KSK rows
    s0 = sk.key_lv0.astype(np.uint32)
    for (i, j, k) in ((0, 0, 1), (5, 3, 2), (1023, 8, 3), (512, 4, 1)):
        row = gk.key_switching_key[i, j, k]
        ph = np.uint32(row[P.n] - (row[:P.n] * s0).sum(dtype=np.uint32))
        want = O.f64_to_torus(float(k * int(sk.key_lv1[i])) / float(1 << ((j + 1) * P.basebit)))
        err = np.int32(np.uint32(ph - np.uint32(want))) / 2.0**32
        assert abs(err) < 8 * P.alpha_lv0, (i, j, k, err)
        assert not gk.key_switching_key[i, j, 0].any()
    masks = gk.key_switching_key[3, 2, 1, :P.n].astype(np.float64)
    assert 0.4 < masks.mean() / 2.0**32 < 0.6  # uniform mask words
    # functional: gates with the generated key
    A = np.array([1, 1, 0, 0, 1, 0], bool)
    B = np.array([1, 0, 1, 0, 1, 1], bool)
    ca, cb = sk.encrypt_bool(A, 77), sk.encrypt_bool(B, 78)
    for op in (O.GATE_NAND, O.GATE_XOR, O.GATE_AND):
        got = eng.batch_gate(op, ca, cb)
        assert np.array_equal(sk.decrypt_bool(got), np.array([O.GATE_TRUTH[op](bool(a), bool(b)) for a, b in zip(A, B)]))
    # the generated key is an ordinary key: the CPU path with it gives the same ciphertexts
    class _K:
        pass
    ock = O.CloudKey.__new__(O.CloudKey)
    ock.params = O.SECURITY_128_BIT
    ock.decomposition_offset = gk.decomposition_offset
    ock.blind_rotate_testvec = gk.blind_rotate_testvec
    ock.bootstrapping_key = gk.bootstrapping_key
    ock.bootstrapping_key_time = None
    ock.key_switching_key = gk.key_switching_key
    assert np.array_equal(eng.batch_gate(O.GATE_NAND, ca, cb), O.batch_gate(ock, O.GATE_NAND, ca, cb))
    # determinism in the seed
    eng2 = R.Engine(P, 0)
    eng2.gen_cloud_key(sk.key_lv0, sk.key_lv1, seed=2025)
    g2 = eng2.export_cloud_key()
    assert np.array_equal(g2.bootstrapping_key, gk.bootstrapping_key) and np.array_equal(g2.key_switching_key, gk.key_switching_key)
    eng2.gen_cloud_key(sk.key_lv0, sk.key_lv1, seed=2026)
    assert not np.array_equal(eng2.export_cloud_key().key_switching_key, gk.key_switching_key)
    eng.close()
    eng2.close()


def _chacha20_block(key_words, counter, nonce_words):
    """RFC 8439 section 2.3 block function (pure Python), the checker for the device keystream."""
    def rotl(x, r):
        return ((x << r) | (x >> (32 - r))) & 0xFFFFFFFF

    st = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(key_words) + [counter] + list(nonce_words)
    x = list(st)

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & 0xFFFFFFFF; x[d] = rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & 0xFFFFFFFF; x[b] = rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & 0xFFFFFFFF; x[d] = rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & 0xFFFFFFFF; x[b] = rotl(x[b] ^ x[c], 7)

    for _ in range(10):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return [(a + b) & 0xFFFFFFFF for a, b in zip(x, st)]


@pytest.mark.gpu
def test_gpu_key_generation_randomness(O, keys128):
    """The generator behind tfhe_hip_gen_cloud_key_{secure,with_key}: mask words are the RFC 8439 ChaCha20
    keystream under the 256-bit generator key (checked against the RFC's own test vector and a pure-Python block
    function), the OS-keyed default gives a different key every call, and keys made either way are ordinary keys."""
    import rs_tfhe_amd as R

    # RFC 8439 section 2.3.2 test vector pins the checker itself
    kw = [int.from_bytes(bytes(range(4 * i, 4 * i + 4)), "little") for i in range(8)]
    blk = _chacha20_block(kw, 1, [0x09000000, 0x4A000000, 0x00000000])
    assert blk[0] == 0xE4E7F110 and blk[15] == 0x4E3C50A2

    sk, _ = keys128
    P = R.params.SECURITY_128_BIT
    eng = R.Engine(P, 0)
    rng_key = bytes(range(100, 132))
    eng.gen_cloud_key(sk.key_lv0, sk.key_lv1, rng_key=rng_key)
    k1 = eng.export_cloud_key()
    words = [int.from_bytes(rng_key[4 * i:4 * i + 4], "little") for i in range(8)]
    base = P.base
    for (i, j, k) in ((0, 0, 1), (7, 2, 3), (1023, P.iks_t - 1, base - 1)):
        row = base * P.iks_t * i + base * j + k
        for x16 in (0, 5, 43):  # mask words 16*x16 .. 16*x16+15 of the row = keystream block x16, nonce (row, 0, "KSK")
            want = _chacha20_block(words, x16, [row, 0, 0x4B534B])
            got = k1.key_switching_key[i, j, k, 16 * x16:min(16 * x16 + 16, P.n)]  # word n is the body, not a mask word
            assert got.tolist() == want[:len(got)], (i, j, k, x16)
    eng.gen_cloud_key(sk.key_lv0, sk.key_lv1, rng_key=rng_key)
    assert np.array_equal(eng.export_cloud_key().key_switching_key, k1.key_switching_key)  # the generator key fixes the key
    # default: keyed by getrandom(2) -- two calls, two keys; and it works as a key
    eng.gen_cloud_key(sk.key_lv0, sk.key_lv1)
    s1 = eng.export_cloud_key()
    eng.gen_cloud_key(sk.key_lv0, sk.key_lv1)
    s2 = eng.export_cloud_key()
    assert not np.array_equal(s1.key_switching_key, s2.key_switching_key)
    assert not np.array_equal(s1.bootstrapping_key, s2.bootstrapping_key)
    assert not np.array_equal(s1.key_switching_key, k1.key_switching_key)
    A = np.array([1, 1, 0, 0, 1, 0, 1], bool)
    B = np.array([1, 0, 1, 0, 1, 1, 0], bool)
    ca, cb = sk.encrypt_bool(A, 5), sk.encrypt_bool(B, 6)
    assert np.array_equal(sk.decrypt_bool(eng.batch_gate(O.GATE_NAND, ca, cb)), ~(A & B))
    with pytest.raises(ValueError):
        eng.gen_cloud_key(sk.key_lv0, sk.key_lv1, rng_key=b"short")
    eng.close()


# ---- several devices behind one handle --------------------------------------------------------------
def test_pool_two_contexts_bit_exact_and_order_preserving(O, eng128, keys128):
    """tfhe_hip_pool with devices = {0, 0} (two contexts on the one GPU of the test box; the driver's 8-GPU run
    uses {0..7}): the key is uploaded once and replicated device to device, every batch call splits the host
    arrays contiguously over the members (rayon_impl.rs:40-47 keeps input order) -- results must equal the
    single-context ones word for word at ragged counts, including counts below the member count."""
    import rs_tfhe_amd as R

    sk, ck = keys128
    pk = _cloud_key(ck)
    P = pk.params
    pool = R.Pool(P, [0, 0])
    assert len(pool) == 2 and pool.members_for(5) == 1 and pool.shard(5, 0) == (0, 5) and pool.shard(5, 1) == (0, 0)
    assert pool.members_for(600) == 2 and pool.shard(601, 0) == (0, 301) and pool.shard(601, 1) == (301, 601)
    pool.load_cloud_key(pk)
    k0, k1 = pool.export_cloud_key(0), pool.export_cloud_key(1)
    assert np.array_equal(k1.bootstrapping_key, k0.bootstrapping_key) and np.array_equal(k1.key_switching_key, k0.key_switching_key)
    assert np.array_equal(k1.key_switching_key, pk.key_switching_key) and k1.decomposition_offset == pk.decomposition_offset
    rng = np.random.default_rng(77)
    for count in (1, 5, 300, 513, 701):  # <= 256: one member; 300 -> 150 + 150; 513 -> 257 + 256; 701 -> 351 + 350
        A, B, Cc = (rng.integers(0, 2, count).astype(bool) for _ in range(3))
        ca, cb, cc = sk.encrypt_bool(A, 100 + count), sk.encrypt_bool(B, 200 + count), sk.encrypt_bool(Cc, 300 + count)
        got = pool.batch_gate(O.GATE_NAND, ca, cb)
        assert np.array_equal(got, eng128.batch_gate(O.GATE_NAND, ca, cb)), count
        assert np.array_equal(sk.decrypt_bool(got), ~(A & B))
        codes = rng.integers(0, 11, count).astype(np.uint8)
        assert np.array_equal(pool.batch_gates_mixed(codes, ca, cb), eng128.batch_gates_mixed(codes, ca, cb))
        assert np.array_equal(pool.batch_bootstrap(ca), eng128.batch_bootstrap(ca))
        tv = rng.integers(0, 2**32, (count, 2, N), dtype=np.uint64).astype(np.uint32)  # per-ciphertext test vectors follow their shard
        assert np.array_equal(pool.batch_bootstrap(ca, tv), eng128.batch_bootstrap(ca, tv))
        assert np.array_equal(pool.batch_bootstrap(ca, tv[0], keyswitch=False), eng128.batch_bootstrap(ca, tv[0], keyswitch=False))
        assert np.array_equal(pool.batch_mux(ca, cb, cc, naive=True), eng128.batch_mux(ca, cb, cc, naive=True))
        assert np.array_equal(pool.batch_mux(ca, cb, cc, naive=False), eng128.batch_mux(ca, cb, cc, naive=False))
        assert np.array_equal(pool.batch_blind_rotate(ca), eng128.batch_blind_rotate(ca))
    assert pool.batch_gate(O.GATE_NAND, np.empty((0, P.n + 1), np.uint32), np.empty((0, P.n + 1), np.uint32)).shape == (0, P.n + 1)
    # key generation through the pool: every member holds the same generated key
    pool.gen_cloud_key(sk.key_lv0, sk.key_lv1, seed=5)
    g0, g1 = pool.export_cloud_key(0), pool.export_cloud_key(1)
    assert np.array_equal(g0.key_switching_key, g1.key_switching_key) and np.array_equal(g0.bootstrapping_key, g1.bootstrapping_key)
    A = rng.integers(0, 2, 9).astype(bool)
    ca = sk.encrypt_bool(A, 1)
    assert np.array_equal(sk.decrypt_bool(pool.batch_gate(O.GATE_NAND, ca, ca)), ~A)
    with pytest.raises(R._capi.TfheHipError):
        pool.batch_gate(99, ca, ca)
    # a member context borrowed for the device-resident entry points (what bench.py times)
    import torch

    member = R.Engine.from_pool(pool, 1)
    ta = torch.from_numpy(ca.view(np.int32)).to("cuda:0")
    to = torch.empty_like(ta)
    member.batch_gate_dev(O.GATE_NAND, ta, ta, to)
    torch.cuda.synchronize()
    assert np.array_equal(to.cpu().numpy().view(np.uint32), pool.batch_gate(O.GATE_NAND, ca, ca))
    member.close()  # borrowed: the pool still owns the context
    assert np.array_equal(sk.decrypt_bool(pool.batch_gate(O.GATE_NAND, ca, ca)), ~A)
    with pytest.raises(ValueError):
        R.Engine.from_pool(pool, 2)
    pool.close()
    with pytest.raises(R._capi.TfheHipError):
        R.Pool(P, [0, 4096])  # no such device: create fails as a whole


# ---- golden fixture (no oracle involved) -----------------------------------------------------
def test_golden_toy_instance(golden):
    import rs_tfhe_amd as R
    from rs_tfhe_amd.params import SecurityParams

    g = golden["toy"]
    n, l, bgbit, basebit, t = (int(v) for v in g["params"])
    P = SecurityParams("TOY_N4", 0, n, l, bgbit, basebit, t, 2.0e-5, 2.0e-8)
    pk = R.CloudKey(P, g["bsk"], g["ksk"], int(g["offset"][0]), g["testvec"])
    eng = R.Engine(P, 0)
    eng.load_cloud_key(pk)
    assert np.array_equal(eng.batch_blind_rotate(g["cts"]), g["blind_rotate"])
    assert np.array_equal(eng.batch_sample_extract(g["blind_rotate"]), g["lv1"])
    assert np.array_equal(eng.batch_identity_key_switch(g["lv1"]), g["keyswitch"])
    assert np.array_equal(eng.batch_bootstrap(g["cts"]), g["bootstrap"])
    assert np.array_equal(eng.batch_bootstrap(g["cts"], keyswitch=False), g["bootstrap_noks"])
    for op in range(10):
        assert np.array_equal(eng.batch_gate(op, g["cts"], g["cts2"]), g[f"gate_{op}"])
    assert np.array_equal(eng.batch_bootstrap(g["cts"], g["lut"]), g["bootstrap_lut"])
    c3 = g["cts"][::-1].copy()
    assert np.array_equal(eng.batch_mux(g["cts"], g["cts2"], c3, naive=True), g["mux_naive"])
    assert np.array_equal(eng.batch_mux(g["cts"], g["cts2"], c3, naive=False), g["mux"])
    assert np.array_equal(eng.batch_external_product(g["ep_in"], g["ep_index"]), g["ep_out"])
    eng.close()


# ---- edge cases and error behaviour ------------------------------------------------------------
def test_device_count_and_all_devices_pool(O, eng128, keys128):
    """`tfhe_hip_device_count` is what a host binding builds "all GPUs of the node" from (rust `default_engine()`,
    the counterpart of Rayon's one-worker-per-CPU default, rayon_impl.rs:15-27): a pool over range(count) bootstraps."""
    import rs_tfhe_amd as R

    sk, ck = keys128
    n = R.engine.device_count()
    assert n >= 1 and n == __import__("torch").cuda.device_count()
    pool = R.Pool(eng128.params, list(range(n)))
    pool.load_cloud_key(_cloud_key(ck))
    bits = np.array([1, 0, 1, 1, 0], bool)
    ct = sk.encrypt_bool(bits, 8101)
    out = pool.batch_gate(O.GATE_NAND, ct, ct)
    assert np.array_equal(out, O.batch_gate(ck, O.GATE_NAND, ct, ct)) and np.array_equal(sk.decrypt_bool(out), ~bits)
    pool.close()


def test_edge_cases_and_errors(O, eng128, keys128):
    import rs_tfhe_amd as R
    from rs_tfhe_amd import _capi

    sk, ck = keys128
    empty = np.zeros((0, 701), np.uint32)
    assert eng128.batch_gate(O.GATE_NAND, empty, empty).shape == (0, 701)
    assert eng128.batch_blind_rotate(empty).shape == (0, 2, N)
    with pytest.raises(_capi.TfheHipError) as e:
        eng128.batch_gate(42, sk.encrypt_bool([1], 1), sk.encrypt_bool([0], 2))
    assert e.value.code == _capi.EINVAL
    with pytest.raises(_capi.TfheHipError) as e:
        eng128.batch_external_product(np.zeros((1, 2, N), np.uint32), [700])
    assert e.value.code == _capi.EINVAL
    fresh = R.Engine(R.params.SECURITY_128_BIT, 0)
    with pytest.raises(_capi.TfheHipError) as e:
        fresh.batch_gate(O.GATE_NAND, sk.encrypt_bool([1], 1), sk.encrypt_bool([0], 2))
    assert e.value.code == _capi.ENOKEY
    fresh.close()
    with pytest.raises(_capi.TfheHipError):
        R.Engine(R.params.SECURITY_128_BIT, 9999)
    # COPY ignores the second operand
    ct = sk.encrypt_bool([1, 0], 3)
    assert np.array_equal(eng128.batch_gate(O.GATE_COPY, ct, None), O.batch_bootstrap(ck, ct))
    # shape mistakes are caught on the host, before any pointer reaches the library
    import torch

    with pytest.raises(ValueError):
        eng128.batch_gate(O.GATE_NAND, ct, ct[:1])
    with pytest.raises(ValueError):
        eng128.batch_bootstrap(ct, testvec=np.zeros((2, 512), np.uint32))
    with pytest.raises(ValueError):
        eng128.batch_mux(ct, ct, ct[:1], naive=True)
    dev = torch.device("cuda:0")
    ta = torch.zeros((4, 701), dtype=torch.int32, device=dev)
    with pytest.raises(ValueError):
        eng128.batch_gate_dev(O.GATE_NAND, ta, ta[:3].contiguous(), torch.empty_like(ta))
    with pytest.raises(ValueError):
        eng128.batch_gate_dev(O.GATE_NAND, ta, ta, torch.empty((4, 700), dtype=torch.int32, device=dev))
    with pytest.raises(ValueError):
        eng128.batch_gate_dev(O.GATE_NAND, ta.cpu(), ta, torch.empty_like(ta))
    with pytest.raises(ValueError):
        eng128.batch_blind_rotate_dev(ta, torch.empty((4, 2, 512), dtype=torch.int32, device=dev))


# ---- device-resident path + full-size properties -------------------------------------------------
def test_device_resident_full_batch_properties(O, eng128, keys128):
    """BASELINE configs[1] size: 65,536 hom_nand on 65,536 DISTINCT ciphertext pairs, device-resident: decrypt-correct on
    every ciphertext, the same pair at another batch position gives the same bits, a second run gives the same batch, and
    2,048 outputs spread over the whole batch (first / middle / last workgroups of both kernels) are bit-exact vs the oracle."""
    import torch

    sk, ck = keys128
    B = 65536
    rng = np.random.default_rng(29)
    bits_a = rng.integers(0, 2, B).astype(bool)
    bits_b = rng.integers(0, 2, B).astype(bool)
    ca, cb = sk.encrypt_bool(bits_a, 601), sk.encrypt_bool(bits_b, 602)
    dup = np.array([0, 3, 4099, 33333])
    for arr, bits in ((ca, bits_a), (cb, bits_b)):
        arr[B - len(dup):] = arr[dup]
        bits[B - len(dup):] = bits[dup]
    dev = torch.device("cuda:0")
    ta = torch.from_numpy(ca.view(np.int32)).to(dev)
    tb = torch.from_numpy(cb.view(np.int32)).to(dev)
    to, to2 = torch.empty_like(ta), torch.empty_like(ta)
    eng128.batch_gate_dev(O.GATE_NAND, ta, tb, to)
    eng128.batch_gate_dev(O.GATE_NAND, ta, tb, to2)
    torch.cuda.synchronize()
    assert torch.equal(to, to2)
    out = to.cpu().numpy().view(np.uint32)
    assert np.array_equal(out[B - len(dup):], out[dup])
    # vectorised decrypt of the whole batch
    phase = out[:, 700] - (out[:, :700] * sk.key_lv0[None, :]).sum(axis=1, dtype=np.uint32)
    assert np.array_equal(phase.view(np.int32) >= 0, ~(bits_a & bits_b))
    idx = np.unique(np.concatenate([np.arange(64), np.linspace(0, B - 1, 1920).astype(np.int64), np.arange(B - 64, B)]))
    assert len(idx) >= 2000
    ref = O.batch_gate(ck, O.GATE_NAND, ca[idx], cb[idx])
    assert np.array_equal(out[idx], ref)
    kt = eng128.kernel_times()
    assert kt["bootstraps"] >= 0


def test_calls_on_different_streams_share_one_context(O, eng128, keys128):
    """Back-to-back asynchronous calls on different streams (torch default, a side stream, the
    context's own stream via the host API) reuse the context's intermediate buffers; none may
    clobber another's in-flight data (include/tfhe_hip.h, stream semantics)."""
    import torch

    sk, ck = keys128
    rng = np.random.default_rng(31)
    B = 3000  # large enough that each call is still running when the next is issued
    dev = torch.device("cuda:0")
    sets = []
    for i in range(3):
        A = rng.integers(0, 2, B).astype(bool)
        Bb = rng.integers(0, 2, B).astype(bool)
        ca, cb = sk.encrypt_bool(A, 3100 + 2 * i), sk.encrypt_bool(Bb, 3101 + 2 * i)
        sets.append((A, Bb, ca, cb))
    t = [(torch.from_numpy(ca.view(np.int32)).to(dev), torch.from_numpy(cb.view(np.int32)).to(dev)) for _, _, ca, cb in sets]
    outs = [torch.empty_like(x[0]) for x in t]
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    eng128.batch_gate_dev(O.GATE_NAND, t[0][0], t[0][1], outs[0])  # torch default stream
    with torch.cuda.stream(side):
        eng128.batch_gate_dev(O.GATE_XOR, t[1][0], t[1][1], outs[1])  # side stream
    host_out = eng128.batch_gate(O.GATE_OR, sets[2][2], sets[2][3])  # context's own stream, blocking
    torch.cuda.synchronize()
    got = [outs[0].cpu().numpy().view(np.uint32), outs[1].cpu().numpy().view(np.uint32), host_out]
    truth = [~(sets[0][0] & sets[0][1]), sets[1][0] ^ sets[1][1], sets[2][0] | sets[2][1]]
    for g, w in zip(got, truth):
        assert np.array_equal(sk.decrypt_bool(g), w)
    for g, gate, (_, _, ca, cb) in zip(got, (O.GATE_NAND, O.GATE_XOR, O.GATE_OR), sets):
        assert np.array_equal(g[:16], O.batch_gate(ck, gate, ca[:16], cb[:16]))


def test_rccl_backend_single_rank(O, keys128):
    """The one-process-per-GPU plumbing under the real "nccl" (= RCCL) backend, world size 1 (the
    box has one GPU): key replication through device memory, scatter / sharded gate / gather on
    device tensors, barrier and the max-over-ranks reduction bench.py uses."""
    import socket

    import torch
    import torch.distributed as dist

    import rs_tfhe_amd as R
    from rs_tfhe_amd import distributed as D

    sk, ck = keys128
    pk = _cloud_key(ck)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        rep = D.broadcast_cloud_key(pk, pk.params, src=0)
        assert np.array_equal(rep.bootstrapping_key, pk.bootstrapping_key)
        assert np.array_equal(rep.key_switching_key, pk.key_switching_key)
        assert rep.decomposition_offset == pk.decomposition_offset
        eng = R.Engine(pk.params, 0)
        eng.load_cloud_key(rep)
        A = np.array([0, 0, 1, 1, 1, 0, 1], bool)
        B = np.array([0, 1, 0, 1, 1, 1, 0], bool)
        ca, cb = sk.encrypt_bool(A, 3300), sk.encrypt_bool(B, 3301)
        fa = torch.from_numpy(ca.view(np.int32)).to(dev)
        fb = torch.from_numpy(cb.view(np.int32)).to(dev)
        sa = D.scatter_batch(fa, len(A), 701, src=0, device=dev)
        sb = D.scatter_batch(fb, len(A), 701, src=0, device=dev)
        so = torch.empty_like(sa)
        D.sharded_batch_gate(eng, O.GATE_NAND, sa, sb, so)
        full = D.gather_batch(so, len(A), 701, dst=0, device=dev)
        dist.barrier()
        t = torch.tensor([1.5], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        torch.cuda.synchronize()
        assert float(t) == 1.5
        got = full.cpu().numpy().view(np.uint32)
        assert np.array_equal(got, O.batch_gate(ck, O.GATE_NAND, ca, cb))
        assert np.array_equal(sk.decrypt_bool(got), ~(A & B))
        # device-to-device key replication in the engine layouts (what bench.py's multi-rank path does): the
        # broadcast runs in place on the aliased key buffers; a second context then receives them by a plain
        # device copy + adopt, and must compute the same words
        D.broadcast_engine_key(eng, src=0)
        eng2 = R.Engine(pk.params, 0)
        src_t, dst_t = eng.cloud_key_device_tensors(), eng2.cloud_key_device_tensors()
        assert src_t[0].numel() == pk.params.bsk_bytes and src_t[0].data_ptr() != dst_t[0].data_ptr()
        for a_, b_ in zip(src_t[:3], dst_t[:3]):
            b_.copy_(a_)
        torch.cuda.synchronize()
        eng2.adopt_cloud_key(src_t[3])
        assert np.array_equal(eng2.batch_gate(O.GATE_NAND, ca, cb), got)
        k2 = eng2.export_cloud_key()
        assert np.array_equal(k2.key_switching_key, pk.key_switching_key) and k2.decomposition_offset == pk.decomposition_offset
        eng3 = R.Engine(pk.params, 0)
        with pytest.raises(R._capi.TfheHipError):
            eng3.adopt_cloud_key(0)  # no buffers yet
        eng.close()
        eng2.close()
        eng3.close()
    finally:
        dist.destroy_process_group()


def test_client_keygen_encrypt_gate_decrypt_round_trip(O):
    """The reference's README flow with the product package alone: SecretKey::new, CloudKey::new,
    encrypt, a gate, decrypt (README.md:60-80 of the reference) -- then the same ciphertexts through
    the CPU path under the exported key must give the same words."""
    import rs_tfhe_amd as R

    P = R.params.SECURITY_128_BIT
    sk = R.SecretKey.new(P, seed=41)
    ck = R.CloudKey.new(sk, seed=42)
    A = np.array([0, 0, 1, 1], bool)
    B = np.array([0, 1, 0, 1], bool)
    ca, cb = sk.encrypt_bool(A, seed=43), sk.encrypt_bool(B, seed=44)
    g = R.Gates()
    out = np.stack([g.nand(x, y, ck) for x, y in zip(ca, cb)])
    assert np.array_equal(sk.decrypt_bool(out), ~(A & B))
    assert np.array_equal(sk.decrypt_bool(R.gates.batch_xor(ca, cb, ck)), A ^ B)
    # CPU path under the very same (GPU-generated, exported) key
    ock = O.CloudKey.from_arrays(O.SECURITY_128_BIT, ck.bootstrapping_key, ck.key_switching_key,
                                 ck.decomposition_offset, ck.blind_rotate_testvec)
    assert np.array_equal(out, O.batch_gate(ock, O.GATE_NAND, ca, cb))
    # programmable bootstrap with the client's message encoding
    P4 = R.params.SECURITY_UINT4
    sk4 = R.SecretKey.new(P4, seed=45)
    ck4 = R.CloudKey.new(sk4, seed=46)
    msgs = np.arange(16)
    lut = R.lut.Generator(16).generate_lookup_table(lambda x: (5 * x + 3) % 16)
    res = R.LutBootstrap().bootstrap_lut(sk4.encrypt_lwe_message(msgs, 16, seed=47), lut, ck4)
    assert np.array_equal(sk4.decrypt_lwe_message(res, 16), (5 * msgs + 3) % 16)


def test_tlwe_lincomb_and_fused_bootstrap(O, eng128, keys128):
    """TLWE Add / Sub / Neg / AddMul / SubMul (tlwe.rs:129-214) word for word, and the fused
    lincomb + bootstrap against 'combine on the host, then bootstrap' through the CPU path."""
    import rs_tfhe_amd as R

    sk, ck = keys128
    rng = np.random.default_rng(51)
    a = rng.integers(0, 2**32, (9, 701), dtype=np.uint64).astype(np.uint32)
    b = rng.integers(0, 2**32, (9, 701), dtype=np.uint64).astype(np.uint32)
    M = 0xFFFFFFFF
    assert np.array_equal(eng128.batch_tlwe_lincomb(1, a, 1, b), a + b)                      # Add
    assert np.array_equal(eng128.batch_tlwe_lincomb(1, a, -1 & M, b), a - b)                 # Sub
    assert np.array_equal(eng128.batch_tlwe_lincomb(-1 & M, a), (0 - a).astype(np.uint32))   # Neg
    assert np.array_equal(eng128.batch_tlwe_lincomb(1, a, 12345, b), a + b * np.uint32(12345))        # AddMul
    assert np.array_equal(eng128.batch_tlwe_lincomb(1, a, -12345 & M, b), a - b * np.uint32(12345))   # SubMul
    exp = a + b
    exp[:, -1] += np.uint32(0x20000000)
    assert np.array_equal(eng128.batch_tlwe_lincomb(1, a, 1, b, 0x20000000), exp)
    with pytest.raises(ValueError):
        eng128.batch_tlwe_lincomb(1, a, 1, None)
    # fused: every gate is a lincomb + bootstrap with the default test vector
    A = np.array([0, 0, 1, 1], bool)
    B = np.array([0, 1, 0, 1], bool)
    ca, cb = sk.encrypt_bool(A, 5100), sk.encrypt_bool(B, 5101)
    got = eng128.batch_lincomb_bootstrap(-1 & M, ca, -1 & M, cb, 0x20000000)
    assert np.array_equal(got, O.batch_gate(ck, O.GATE_NAND, ca, cb))
    # and with a lookup table: PBS(x + y) == CPU path on the host-side sum
    msgs_x, msgs_y = np.array([1, 0, 1, 0]), np.array([0, 0, 1, 1])
    cx, cy = sk.encrypt_lwe_message(msgs_x, 4, 5102), sk.encrypt_lwe_message(msgs_y, 4, 5103)
    lut = R.lut.Generator(4).generate_lookup_table(lambda v: (3 * v + 1) % 4)
    fused = eng128.batch_lincomb_bootstrap(1, cx, 1, cy, testvec=lut.poly)
    assert np.array_equal(fused, O.batch_bootstrap(ck, cx + cy, testvec=lut.poly))
    assert np.array_equal(sk.decrypt_lwe_message(fused, 4), (3 * (msgs_x + msgs_y) + 1) % 4)
    assert np.array_equal(eng128.batch_lincomb_bootstrap(1, cx, 1, cy, testvec=lut.poly, keyswitch=False),
                          O.batch_bootstrap(ck, cx + cy, testvec=lut.poly, keyswitch=False))


def test_lut_nibble_adder_circuit(O, eng128, keys128, keys_uint4):
    """examples/lut_add_two_numbers.rs for a batch of byte pairs: three programmable bootstraps per
    pair, device-resident.
    (1) SECURITY_128_BIT, the set the example names: bit-identical to the same composition on the CPU
        path.  No decrypt claim there: the reference's truncating decomposition (trgsw.rs:144-171, no
        rounding term in key.rs:78-89) leaves a deterministic error of std 0.0095 on every bootstrap
        output at bgbit = 6 (measured on the CPU path with all key noise set to zero), more than the
        1/128 half-step of modulus 32, so ~40 % of nibbles mis-decode on the CPU path too.
    (2) SECURITY_UINT4 (bgbit = 22): the decrypted bytes are the plain sums."""
    import torch

    import rs_tfhe_amd as R
    from rs_tfhe_amd import circuit

    rng = np.random.default_rng(52)
    a = np.concatenate([[42, 255, 0, 15], rng.integers(0, 256, 28)]).astype(np.int64)
    b = np.concatenate([[137, 255, 0, 1], rng.integers(0, 256, 28)]).astype(np.int64)
    gen = R.lut.Generator(32)
    mod16 = gen.generate_lookup_table(lambda x: x % 16).poly
    carry = gen.generate_lookup_table(lambda x: 1 if x >= 16 else 0).poly
    dev = torch.device("cuda:0")

    def run(eng, sk, seed):
        enc = lambda v, s: sk.encrypt_lwe_message(v, 32, s)  # noqa: E731
        cts = enc(a & 15, seed), enc(a >> 4, seed + 1), enc(b & 15, seed + 2), enc(b >> 4, seed + 3)
        t = [torch.from_numpy(x.view(np.int32)).to(dev) for x in cts]
        out = circuit.lut_add_u8_dev(eng, *t)
        torch.cuda.synchronize()
        out = tuple(x.cpu().numpy().view(np.uint32) for x in out)
        host = circuit.lut_add_u8(eng, *cts)  # the host-array form gives the same words
        assert all(np.array_equal(x, y) for x, y in zip(out, host))
        return cts, out

    # (1) word-for-word against the CPU path
    sk, ck = keys128
    (al, ah, bl, bh), (sl, sh, cr) = run(eng128, sk, 5200)
    o_sl = O.batch_bootstrap(ck, al + bl, testvec=mod16)
    o_cr = O.batch_bootstrap(ck, al + bl, testvec=carry)
    o_sh = O.batch_bootstrap(ck, ah + bh + o_cr, testvec=mod16)
    assert np.array_equal(sl, o_sl) and np.array_equal(cr, o_cr) and np.array_equal(sh, o_sh)

    # (2) correct bytes where the parameter set has the precision for modulus 32
    sk4, ck4 = keys_uint4
    pk4 = _cloud_key(ck4)
    eng4 = R.bootstrap.engine_for(pk4.params, 0)
    eng4.ensure_key(pk4)
    _, (sl, sh, cr) = run(eng4, sk4, 5300)
    res = sk4.decrypt_lwe_message(sl, 32) | (sk4.decrypt_lwe_message(sh, 32) << 4)
    want = (a + b) & 0xFF
    assert res[0] == 179  # the example's own 42 + 137
    # n = 820 mask words rounded to 2N positions leave ~2.7 sigma per bootstrap at modulus 32
    assert (res == want).mean() >= 0.9
    assert np.array_equal(sk4.decrypt_lwe_message(cr, 32)[res == want], ((a & 15) + (b & 15) >= 16)[res == want])


def test_one_context_from_several_threads(O, eng128, keys128):
    """`Bootstrap: Send + Sync` (src/bootstrap/mod.rs:23): one context called concurrently from several
    host threads (ctypes releases the GIL for the duration of a call) must serialise internally and
    give every caller its own, correct result."""
    import threading

    sk, ck = keys128
    rng = np.random.default_rng(61)
    jobs = []
    for t in range(6):
        n = int(rng.integers(1, 40))
        A = rng.integers(0, 2, n).astype(bool)
        B = rng.integers(0, 2, n).astype(bool)
        jobs.append((t % 10, A, B, sk.encrypt_bool(A, 6100 + 2 * t), sk.encrypt_bool(B, 6101 + 2 * t)))
    results = [None] * len(jobs)
    errors = []

    def work(i):
        try:
            gate, _, _, ca, cb = jobs[i]
            for _ in range(3):  # repeated, interleaved with the other threads
                results[i] = eng128.batch_gate(gate, ca, cb)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors
    for (gate, A, B, ca, cb), got in zip(jobs, results):
        assert np.array_equal(got, O.batch_gate(ck, gate, ca, cb))


# ---- the N > 1 paths of bench.py, run here so that their first execution is not the driver's 8-GPU box ----------
def _run_bench(extra_args, env_extra, timeout=900):
    """bench.py as a FRESH child process (its own HIP runtime), the one JSON line it prints."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.update(env_extra)
    env.pop("MASTER_PORT", None)  # bench.py picks a free port for its torchrun child
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + extra_args, cwd=root, env=env,
                       capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, (p.returncode, p.stdout[-2000:], p.stderr[-4000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]  # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_bench_two_ranks_share_one_gpu():
    """`bench.py --gpus 2` end to end with two ranks on this box's one GPU (BENCH_SHARE_GPU=1: rendezvous over gloo,
    since RCCL refuses two ranks per device): torchrun child, key generated on rank 0 and broadcast in the engine
    layouts, every rank bootstraps its own shard, barrier + max-over-ranks timing, rank 0 prints the line."""
    d = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "2048", "--no-cpu-baseline"],
                   {"BENCH_SHARE_GPU": "1"})
    assert d["n_gpus"] == 2 and d["decrypt_ok"] is True and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 4096 and d["value"] > 0 and d["unit"] == "bootstraps/s"
    assert d["roofline"]["avg_launch_ms"] > 0 and d["cpu_baseline"] is None


def test_bench_two_ranks_mixed_circuit_80bit():
    """The same with BASELINE configs[4]'s gate mix (half Gates::mux, half hom_xor) at SECURITY_80_BIT."""
    d = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "2048", "--no-cpu-baseline",
                    "--gate", "mixed", "--params", "SECURITY_80_BIT"], {"BENCH_SHARE_GPU": "1"})
    assert d["n_gpus"] == 2 and d["decrypt_ok"] is True and "SECURITY_80_BIT" in d["metric"]


def test_bench_eight_ranks_share_one_gpu():
    """`bench.py --gpus 8`, the N = 8 branch the driver's scaling sweep runs, end to end on this box's one GPU (eight
    ranks, BENCH_SHARE_GPU=1): no MASTER_PORT given, so the torchrun child gets a port that is free now; the line carries
    every rank's own time for the steps, its kernels' launch times and the key broadcast's duration."""
    for extra in ([], ["--gate", "mixed", "--params", "SECURITY_80_BIT"]):
        d = _run_bench(["--gpus", "8", "--steps", "1", "--warmup", "0", "--batch", "1024", "--no-cpu-baseline"] + extra,
                       {"BENCH_SHARE_GPU": "1"})
        assert d["n_gpus"] == 8 and d["decrypt_ok"] is True and d["scaling"] == "weak"
        assert d["config"]["global_batch"] == 8192 and d["value"] > 0
        pr = d["per_rank"]
        assert all(len(pr[k]) == 8 for k in ("ms_per_step", "blind_rotate_ms", "key_switch_ms", "shader_mhz", "key_broadcast_s"))
        assert d["ms_per_step_max_rank"] == max(pr["ms_per_step"]) and d["ms_per_step_min_rank"] > 0
        assert d["ms_per_step"] >= d["ms_per_step_max_rank"] - 0.01  # the line's time is the max over ranks
        assert d["key_broadcast_s"] > 0 and d["key_broadcast_backend"] == "gloo"
        # BASELINE configs[2] as one caller has it: ONE pool handle over the eight members, the GLOBAL batch resident on
        # member 0's GPU (a fresh child of rank 0, after the ranks have let go of the GPU); here the members share a
        # device, so the shards move by peer copies
        pr = d["pool_resident"]
        assert "error" not in pr, pr
        assert pr["batch_total"] == 8192 and len(pr["devices"]) == 8 and pr["transport"] == "peer-copy" and pr["value"] > 0
        if not extra:
            assert pr["decrypt_ok"] is True and pr["oracle_sample_equal"] is True and pr["oracle_sample"] == 16


def test_bench_line_carries_the_other_configs_and_the_concurrent_callers():
    """The N = 1 line (what the driver records as BENCH_rNN) beside its headline: short runs of BASELINE configs[3] and
    of configs[4]'s single-GPU share, the host-buffer path pageable and pinned, and the aggregate rate of a team of
    threads making one-ciphertext calls (cpu_baseline.gpu_concurrent_single_gate)."""
    d = _run_bench(["--steps", "2", "--warmup", "1", "--batch", "4096", "--cpu-seconds", "3"], {}, timeout=1200)
    oc = d["other_configs"]
    for k in ("configs3_pbs_uint4", "configs4_share_mixed_80bit", "host_path_pageable", "host_path_pinned"):
        assert "error" not in oc[k], oc[k]
        assert oc[k]["value"] > 0 and oc[k]["decrypt_ok"] is True, oc[k]
    assert "SECURITY_UINT4" in oc["configs3_pbs_uint4"]["metric"] and 0 < oc["configs3_pbs_uint4"]["roofline_frac"] < 1
    assert "SECURITY_80_BIT" in oc["configs4_share_mixed_80bit"]["metric"]
    cc = d["cpu_baseline"]["gpu_concurrent_single_gate"]
    assert cc["threads"] == [1, 8, 64, 256] and len(cc["gates_per_s"]) == 4
    assert cc["gates_per_s"][2] > 10 * cc["unmerged_gates_per_s_8_threads"], cc
    assert d["cpu_baseline"]["gpu_matches_cpu_bit_exact"] is True and d["decrypt_ok"] is True


def test_bench_pool_two_members_pinned_and_pageable():
    """`bench.py --pool-devices 0,0`: ONE process, the multi-device handle a Rust caller binds (tfhe_hip_pool_*), two
    member contexts on this GPU, host buffers -- pinned (zero-copy) and pageable (each member stages its shard
    through its own pinned arena)."""
    for extra in (["--pinned"], []):
        d = _run_bench(["--pool-devices", "0,0", "--steps", "1", "--warmup", "1", "--batch", "2048"] + extra, {})
        assert d["devices"] == [0, 0] and d["decrypt_ok"] is True and d["batch_total"] == 4096
        assert d["host_memory"].startswith("pinned" if extra else "pageable")


def test_single_gate_latency_warm_and_after_idle(O, eng128, keys128):
    """BASELINE configs[0] through the GPU: ONE `Gates::nand`-shaped call with host buffers (what criterion's `gate_nand`
    times on the CPU, benches/gate_benchmarks.rs:12-20), as bench.py measures it -- median of back-to-back calls, and of
    calls that each follow an idle gap.  Measured on MI355X boxes: 2.18 ms back to back (2.15 of it in the two kernels),
    2.2 after 10 ms, 2.3 after 1 s (the kernels run at 2.16: the extra is the host side waking up), 2.7 after 10 s (the
    kernels 2.56: the shader clock ramps).  The bounds leave room for the slowest box seen (+6 %) and a noisy host: this
    suite runs with -x, and a timing assertion must not be what stops it."""
    import bench

    sk, ck = keys128
    ca = sk.encrypt_bool(np.array([1, 0, 1, 1, 0, 0, 1, 0], bool), 7001)
    cb = sk.encrypt_bool(np.array([1, 1, 0, 1, 0, 1, 0, 0], bool), 7002)
    lat = bench.single_gate_latency(eng128, O.GATE_NAND, ca, cb, schedule=((0.0, 50), (0.010, 20), (1.0, 3)))
    warm, idle10ms, idle1s = lat["0s"], lat["0.01s"], lat["1s"]
    assert np.array_equal(eng128.batch_gate(O.GATE_NAND, ca, cb), O.batch_gate(ck, O.GATE_NAND, ca, cb))
    # the timing bounds do not gate the suite (a noisy host or a clock ramp is not a wrong result): outside them the test
    # is reported as xfailed with the table, and the run goes on
    slow = []
    if not warm["wall_ms_median"] < 3.0:  # (round 4's committed line read 9.48 ms)
        slow.append("warm median >= 3.0 ms")
    if not warm["wall_ms_median"] - warm["kernels_ms_median"] < 0.4:  # copies + launches + the synchronise: 0.03 measured
        slow.append("host side of a warm call >= 0.4 ms")
    if not (idle10ms["wall_ms_median"] < 3.2 and idle1s["wall_ms_median"] < 5.0):
        slow.append("calls after an idle gap slower than 3.2 / 5.0 ms")
    if slow:
        pytest.xfail(f"single-gate latency outside the expected range on this box ({'; '.join(slow)}): {lat}")
