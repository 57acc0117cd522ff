"""The HIP kernels for sample_extract_index and identity_key_switching held to the closed forms of tests/closed_forms.py
(the same identities tests/test_oracle_closed_forms.py holds the CPU checker to): the extraction against the REFERENCE's
own compiled SPQLIOS product when oracle/_ref travels with the snapshot, the key switch against exact arithmetic under a
noise-free key-switching key."""
import numpy as np
import pytest

import closed_forms as CF
from test_gpu_parity import _cloud_key, _product_params, eng128  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu
N = 1024


def test_gpu_sample_extract_every_index_against_the_reference_product(O, eng128, keys128):
    """k_sample_extract at EVERY k (trlwe.rs:106-120): phase of the extracted sample under key_lv1 == coefficient k of
    b - a (*) s from the reference's Spqlios_poly_mul_1024 + #{i > k : s_i = 1} (quirk Q1)."""
    if not O.ref_available():
        pytest.skip("oracle/_ref not built (it is built from /root/reference and travels as a binary)")
    sk, _ = keys128
    rng = np.random.default_rng(43)
    a = (rng.integers(0, 2**24, (3, N), dtype=np.uint64) * 256).astype(np.uint32)
    b = rng.integers(0, 2**32, (3, N), dtype=np.uint64).astype(np.uint32)
    trlwe = np.stack([a, b], axis=1)  # [3][2][N]
    exp = np.stack([CF.extract_phase_expected(O, t, sk.key_lv1) for t in trlwe])  # [3][N]
    for k in range(N):
        got = CF.lv1_phase(eng128.batch_sample_extract(trlwe, k), sk.key_lv1)
        assert np.array_equal(got, exp[:, k]), k


@pytest.mark.parametrize("setname,kernel", [
    ("SECURITY_128_BIT", "auto"),     # base 4: split kernel at 5, matrix cores at 400
    ("SECURITY_128_BIT", "b4"),
    ("SECURITY_128_BIT", "generic"),
    ("SECURITY_UINT4", "auto"),       # base 32: split at 5, column-sliced at 400
    ("SECURITY_UINT4", "generic"),
    ("SECURITY_UINT7", "auto"),       # base 128, n = 1160
])
def test_gpu_key_switch_exact_phase_under_a_noise_free_key(O, monkeypatch, setname, kernel):
    """Every key-switch kernel family under a key-switching key generated with alpha = 0 (key.rs:102-122): the output's
    phase under key_lv0 must equal src.b - sum_i s1_i * trunc_{t basebit}(a_i + PREC_OFFSET) EXACTLY -- PREC_OFFSET, digit
    order, the row index base*t*i + base*j + k and the sign, none of which a decrypt test sees (trgsw.rs:332-360)."""
    import rs_tfhe_amd as R

    P = O.PARAM_SETS[setname]
    sk = O.SecretKey(P, 78)
    ksk = CF.noise_free_ksk(O, P, sk)
    pp = _product_params(P)
    pk = R.CloudKey(pp, np.zeros((P.n, 2 * P.l, 2, N)), ksk, 0, np.zeros((2, N), np.uint32))
    monkeypatch.setenv("TFHE_HIP_KS_KERNEL", kernel)
    eng = R.Engine(pp, 0)
    eng.load_cloud_key(pk)
    rng = np.random.default_rng(44)
    bits = P.basebit * P.t
    for count in (5, 400):
        lv1 = rng.integers(0, 2**32, (count, N + 1), dtype=np.uint64).astype(np.uint32)
        lv1[0, :N] = 0
        lv1[-1, :N] = 0xFFFFFFFF
        got = sk.phase(eng.batch_identity_key_switch(lv1))
        assert np.array_equal(got, CF.key_switch_phase_expected(P, lv1, sk.key_lv1)), (setname, kernel, count, eng.describe_dispatch(count))
        assert not np.array_equal(got, CF.key_switch_phase_expected(P, lv1, sk.key_lv1, prec_offset=1 << (32 - bits)))
    eng.close()


@pytest.mark.parametrize("setname,m", [("SECURITY_128_BIT", 2), ("SECURITY_128_BIT", 16), ("SECURITY_UINT4", 16), ("SECURITY_80_BIT", 4)])
def test_gpu_trivial_ciphertexts_read_the_table_exactly(O, setname, m):
    """The fused GPU bootstrap (prologue, all n CMUX steps, extraction) without key switch on ciphertexts with a zero mask
    returns the lookup table's entry for the phase EXACTLY, whatever the key (tests/closed_forms.py) -- through the batch
    kernel, the latency kernels and the merged front end alike: b~ (Q2), the rotation's direction, MAX - x on the one slot
    that wraps (Q1), the table's layout, the encoder; and the n CMUX steps of a zero rotation are the identity."""
    import rs_tfhe_amd as R
    from conftest import oracle_keys

    sk, ck = oracle_keys(O, getattr(O, setname), with_time=(setname != "SECURITY_80_BIT"))
    pk = _cloud_key(ck)
    n = pk.params.n
    eng = R.Engine(pk.params, 0)
    eng.load_cloud_key(pk)
    for f in (lambda x: x, lambda x: (x * x + 1) % m, lambda x: (m - 1 - x) % m):
        phases, expect = CF.lut_trivial_cases(f, m)
        lut = R.lut.Generator(m).generate_lookup_table(f).poly
        cts = CF.trivial_ciphertexts(n, phases)
        out = eng.batch_bootstrap(cts, lut, keyswitch=False)  # small: the merged front end, latency kernel
        assert np.array_equal(out[:, n], expect)
        assert np.array_equal(out[:, :n], CF.trivial_mask_expected(n, phases))  # sample_extract_index_2 of the rotated zero mask
        reps = -(-1100 // len(cts))
        big = np.tile(cts, (reps, 1))  # 1,100+: the batch kernel (+ a latency-kernel tail), the direct path
        outb = eng.batch_bootstrap(big, lut, keyswitch=False)
        assert np.array_equal(outb[:, n], np.tile(expect, reps))
        assert np.array_equal(outb[:len(cts), :n], CF.trivial_mask_expected(n, phases))
    phases, expect = CF.gate_testvec_trivial_cases()
    assert np.array_equal(eng.batch_bootstrap(CF.trivial_ciphertexts(n, phases), keyswitch=False)[:, n], expect)
    eng.close()


def test_gpu_gate_prep_on_trivial_inputs_is_the_linear_form_of_gates_rs(O, eng128, keys128):
    """The gate prep fused into the blind rotation's prologue (kGateCa / Cb / Cc) against the linear forms of gates.rs:54-150,
    with no key and no noise: trivial inputs with random phases through tfhe_hip_batch_gates_mixed_nks (per-ciphertext gate
    codes, bootstrap without key switch) -- 400 exact constraints per gate; once merged (small call) and once as a batch."""
    n = eng128.params.n
    rng = np.random.default_rng(46)
    pa, pb = rng.integers(0, 2**32, 4000, dtype=np.uint64), rng.integers(0, 2**32, 4000, dtype=np.uint64)
    codes = (np.arange(4000) % 10).astype(np.uint8)
    a, b = CF.trivial_ciphertexts(n, pa), CF.trivial_ciphertexts(n, pb)
    exp = np.empty(4000, np.uint32)
    for g in range(10):
        exp[codes == g] = CF.gate_trivial_expected(g, pa[codes == g], pb[codes == g])
    out = eng128.batch_gates_mixed(codes, a, b, keyswitch=False)  # 4,000: the batch kernel + a tail
    assert np.array_equal(out[:, n], exp)
    out = eng128.batch_gates_mixed(codes[:200], a[:200], b[:200], keyswitch=False)  # 200: the merged front end
    assert np.array_equal(out[:, n], exp[:200])
    for g in (0, 3, 4):  # and the one-gate entry point with the key switch: the result must at least decrypt like the closed form
        got = eng128.batch_gate(g, a[:64], b[:64])
        want = CF.gate_trivial_expected(g, pa[:64], pb[:64])
        sk, _ = keys128
        assert np.array_equal(sk.decrypt_bool(got), want == 0x20000000), g
