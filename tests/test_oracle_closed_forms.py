"""sample_extract_index and identity_key_switching of the CPU checker held to closed forms (tests/closed_forms.py):
the first against the reference's own compiled SPQLIOS product, the second against exact arithmetic under a noise-free
key-switching key.  CPU only."""
import numpy as np
import pytest

import closed_forms as CF

N = 1024


def test_sample_extract_every_index_against_the_reference_product(O, keys128):
    """trlwe.rs:106-120 at EVERY k: phase of the extracted sample == coefficient k of b - a (*) s computed with the
    reference's Spqlios_poly_mul_1024, plus the count of wrapped positions with s_i = 1 (the MAX - x quirk)."""
    if not O.ref_available():
        pytest.skip("oracle/_ref not built (reference sources absent)")
    sk, _ = keys128
    rng = np.random.default_rng(41)
    for _ in range(2):
        a = (rng.integers(0, 2**24, N, dtype=np.uint64) * 256).astype(np.uint32)
        b = rng.integers(0, 2**32, N, dtype=np.uint64).astype(np.uint32)
        trlwe = np.stack([a, b])
        exp = CF.extract_phase_expected(O, trlwe, sk.key_lv1)
        got = np.array([CF.lv1_phase(O.sample_extract_index(trlwe, k), sk.key_lv1)[0] for k in range(N)], np.uint32)
        assert np.array_equal(got, exp)
        # what the identity pins: true negation instead of MAX - x is off by the wrapped count, the other rotation
        # direction lands far away
        true_neg = (exp - (sk.key_lv1.astype(np.int64).sum() - np.cumsum(sk.key_lv1.astype(np.int64))).astype(np.uint32)).astype(np.uint32)
        assert not np.array_equal(got, true_neg)


@pytest.mark.parametrize("setname,n", [("SECURITY_128_BIT", None), ("SECURITY_UINT4", None), ("SECURITY_UINT7", 96)])
def test_key_switch_exact_phase_under_a_noise_free_key(O, setname, n):
    """trgsw.rs:332-360 for base 4 / 32 / 128 (SECURITY_UINT7's base and t at a reduced n, so that the 1.8 GB key is not
    built on the CPU suite): output phase == src.b - sum_i s1_i * trunc(a_i + PREC_OFFSET), exactly."""
    import dataclasses

    P = O.PARAM_SETS[setname]
    if n:
        P = dataclasses.replace(P, n=n)
    sk = O.SecretKey(P, 77)
    ksk = CF.noise_free_ksk(O, P, sk)
    ck = O.CloudKey.from_arrays(P, np.zeros((P.n, 2 * P.l, 2, N)), ksk, 0, np.zeros((2, N), np.uint32))
    rng = np.random.default_rng(42)
    lv1 = rng.integers(0, 2**32, (6, N + 1), dtype=np.uint64).astype(np.uint32)
    lv1[0, :N] = 0
    lv1[1, :N] = 0xFFFFFFFF
    out = np.stack([O.identity_key_switching(ck, x) for x in lv1])
    got = sk.phase(out)
    assert np.array_equal(got, CF.key_switch_phase_expected(P, lv1, sk.key_lv1))
    # a wrong PREC_OFFSET (one bit too high / too low / absent), fails the identity: the decrypt-style test cannot see this
    bits = P.basebit * P.t
    for wrong in (1 << (32 - bits), 1 << (32 - (2 + bits)), 0):
        assert not np.array_equal(got, CF.key_switch_phase_expected(P, lv1, sk.key_lv1, prec_offset=wrong))


@pytest.mark.parametrize("setname,m", [("SECURITY_128_BIT", 2), ("SECURITY_128_BIT", 16), ("SECURITY_UINT4", 16)])
def test_trivial_ciphertexts_read_the_table_exactly(O, setname, m):
    """The composed bootstrap without key switch (vanilla.rs:54-63, lut.rs:79-99) on ciphertexts with a zero mask returns
    the lookup table's entry for the phase EXACTLY, whatever the key (tests/closed_forms.py): the initial rotation, its
    direction, the MAX - x quirk on the one slot that wraps, the table's layout and the encoder in one identity."""
    from conftest import oracle_keys

    sk, ck = oracle_keys(O, getattr(O, setname), with_time=True)
    n = ck.params.n
    for f in (lambda x: x, lambda x: (x * x + 1) % m, lambda x: (m - 1 - x) % m):
        phases, expect = CF.lut_trivial_cases(f, m)
        out = O.batch_bootstrap(ck, CF.trivial_ciphertexts(n, phases), testvec=O.lut_generate(f, m), keyswitch=False)
        assert np.array_equal(out[:, n], expect)
        assert np.array_equal(out[:, :n], CF.trivial_mask_expected(n, phases))  # sample_extract_index_2 of the rotated zero mask
    phases, expect = CF.gate_testvec_trivial_cases()
    out = O.batch_bootstrap(ck, CF.trivial_ciphertexts(n, phases), keyswitch=False)
    assert np.array_equal(out[:, n], expect)


def test_gate_prep_on_trivial_inputs_is_the_linear_form_of_gates_rs(O, keys128):
    """Every gate's linear prep (gates.rs:54-150) through the composed bootstrap without key switch on trivial inputs with
    random phases: 400 exact constraints per gate on (coefficient of a, coefficient of b, constant), no key involved."""
    sk, ck = keys128
    n = ck.params.n
    rng = np.random.default_rng(45)
    pa, pb = rng.integers(0, 2**32, 400, dtype=np.uint64), rng.integers(0, 2**32, 400, dtype=np.uint64)
    a, b = CF.trivial_ciphertexts(n, pa), CF.trivial_ciphertexts(n, pb)
    for gate in range(10):
        prep = np.stack([O.gate_prep(gate, x, y, n) for x, y in zip(a, b)])
        assert not prep[:, :n].any()
        out = O.batch_bootstrap(ck, prep, keyswitch=False)
        assert np.array_equal(out[:, n], CF.gate_trivial_expected(gate, pa, pb)), gate
