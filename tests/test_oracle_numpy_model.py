"""Pin the C restatement (oracle/tfhe_oracle.c) against a second, independently written numpy
restatement of the same reference functions (oracle/numpy_model.py; numpy's FFT as the 512-point
DFT).  Where the f64 product is exact after rounding (bgbit = 6) the two must agree in every word."""
import numpy as np
import pytest

from oracle import numpy_model as M


def test_stage_functions_agree(O, keys128):
    sk, ck = keys128
    P = ck.params
    rng = np.random.default_rng(71)
    poly = rng.integers(0, 2**32, 1024, dtype=np.uint64).astype(np.uint32)
    # spectra: same layout and values up to f64 round-off of two different FFT implementations
    a, b = M.klemsa_ifft(poly), O.klemsa_ifft(poly)
    assert np.abs(a - b).max() <= 1e-11 * np.abs(b).max()
    assert np.array_equal(M.klemsa_fft(O.klemsa_ifft(poly)), poly)
    assert np.array_equal(O.klemsa_fft(M.klemsa_ifft(poly)), poly)
    for k in (0, 1, 511, 1023, 1024, 1025, 2047, 2048):
        assert np.array_equal(M.poly_mul_with_x_k(poly, k), O.poly_mul_with_x_k(poly, k))
    pa = rng.integers(0, 2**32, 1024, dtype=np.uint64).astype(np.uint32)
    assert np.array_equal(M.decomposition(poly, pa, P.l, P.bgbit, ck.decomposition_offset),
                          O.decomposition(np.stack([poly, pa]), P.l, P.bgbit, ck.decomposition_offset))
    ea, eb = M.external_product(ck.bootstrapping_key[3], poly, pa, P.l, P.bgbit, ck.decomposition_offset)
    ref = O.external_product_fft(ck.bootstrapping_key[3], np.stack([poly, pa]), P.l, P.bgbit, ck.decomposition_offset)
    assert np.array_equal(np.stack([ea, eb]), ref)
    lv1 = rng.integers(0, 2**32, 1025, dtype=np.uint64).astype(np.uint32)
    assert np.array_equal(M.identity_key_switching(lv1, ck.key_switching_key, P.n, P.basebit, P.t),
                          O.identity_key_switching(ck, lv1))
    tr = rng.integers(0, 2**32, (2, 1024), dtype=np.uint64).astype(np.uint32)
    assert np.array_equal(M.sample_extract_index0(tr[0], tr[1]), O.sample_extract_index(tr, 0))


@pytest.mark.parametrize("name,gate_id", [("nand", 0), ("xor", 3), ("xnor", 4)])
def test_whole_gate_bootstrap_agrees_word_for_word(O, keys128, name, gate_id):
    sk, ck = keys128
    P = ck.params
    for a_bit, b_bit, seed in ((True, False, 7100), (True, True, 7102)):
        ca, cb = sk.encrypt_bool([a_bit], seed)[0], sk.encrypt_bool([b_bit], seed + 1)[0]
        got = M.gate(name, ca, cb, ck.bootstrapping_key, ck.key_switching_key, ck.blind_rotate_testvec,
                     P.n, P.l, P.bgbit, P.basebit, P.t, ck.decomposition_offset)
        exp = O.batch_gate(ck, gate_id, ca, cb)[0]
        assert np.array_equal(got, exp)
        assert bool(sk.decrypt_bool(got)[0]) == O.GATE_TRUTH[gate_id](a_bit, b_bit)


def test_blind_rotate_with_lookup_table_agrees(O, keys128):
    sk, ck = keys128
    P = ck.params
    ct = sk.encrypt_lwe_message([1], 2, 7200)[0]
    lut = O.lut_generate(lambda x: 1 - x, 2)
    ra, rb = M.blind_rotate(ct, ck.bootstrapping_key, lut, P.l, P.bgbit, ck.decomposition_offset)
    ref = O.batch_blind_rotate(ck, ct[None], testvec=lut)[0]
    assert np.array_equal(np.stack([ra, rb]), ref)
