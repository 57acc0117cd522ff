"""Client-side recipes of the product package (rs-tfhe_amd/client.py) against the oracle's
restatement of src/tlwe.rs and src/utils.rs.  Integer work: bit-exact wherever both sides see the
same ciphertext; encryption is randomised, so it is checked through decryption and noise statistics."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def C():
    from rs_tfhe_amd import client

    return client


def test_f64_to_torus_matches_reference_cast(O, C):
    xs = np.array([0.0, 0.125, -0.125, 0.25, -0.25, 0.5, -0.5, 0.999999, -0.999999, 1.0, 1.5, -1.5, 2.75, 1e-10, -1e-10,
                   3.0517578125e-05, 1 / 32, 15 / 32, -7 / 32])
    got = C.f64_to_torus(xs)
    exp = np.array([O.f64_to_torus(float(x)) for x in xs], np.uint32)
    assert np.array_equal(got, exp)
    assert got[1] == 0x20000000 and got[2] == 0xE0000000 and got[3] == 0x40000000 and got[4] == 0xC0000000
    rng = np.random.default_rng(1)
    xs = rng.normal(0, 0.3, 2000)
    assert np.array_equal(C.f64_to_torus(xs), np.array([O.f64_to_torus(float(x)) for x in xs], np.uint32))
    ts = rng.integers(0, 2**32, 100, dtype=np.uint64).astype(np.uint32)
    assert np.array_equal(C.torus_to_f64(ts), np.array([O.torus_to_f64(int(t)) for t in ts]))


@pytest.mark.parametrize("setname", ["SECURITY_128_BIT", "SECURITY_80_BIT", "SECURITY_UINT4"])
def test_decrypt_matches_oracle_on_the_same_ciphertexts(O, C, setname):
    import rs_tfhe_amd as R

    op = getattr(O, setname)
    osk = O.SecretKey(op, 77)
    sk = C.SecretKey(R.params.PARAM_SETS[setname], osk.key_lv0, osk.key_lv1)
    rng = np.random.default_rng(2)
    bits = rng.integers(0, 2, 64).astype(bool)
    cts = osk.encrypt_bool(bits, 5)
    assert np.array_equal(sk.phase(cts), osk.phase(cts))
    assert np.array_equal(sk.decrypt_bool(cts), bits)
    assert np.array_equal(sk.decrypt_bool(cts), osk.decrypt_bool(cts))
    for m in (2, 4, 16):
        msgs = rng.integers(0, m, 64)
        cm = osk.encrypt_lwe_message(msgs, m, 6)
        assert np.array_equal(sk.decrypt_lwe_message(cm, m), msgs)
        assert np.array_equal(sk.decrypt_lwe_message(cm, m), osk.decrypt_lwe_message(cm, m))
    # arbitrary words, not just valid encryptions: same decode on both sides
    junk = rng.integers(0, 2**32, (200, op.n + 1), dtype=np.uint64).astype(np.uint32)
    assert np.array_equal(sk.decrypt_bool(junk), osk.decrypt_bool(junk))
    assert np.array_equal(sk.decrypt_lwe_message(junk, 16), osk.decrypt_lwe_message(junk, 16))


def test_encrypt_round_trip_and_noise(O, C):
    import rs_tfhe_amd as R

    P = R.params.SECURITY_128_BIT
    sk = C.SecretKey.new(P, seed=9)
    assert sk.key_lv0.shape == (700,) and sk.key_lv1.shape == (1024,) and set(np.unique(sk.key_lv0)) <= {0, 1}
    osk = O.SecretKey(O.SECURITY_128_BIT, 1)
    osk.key_lv0[:] = sk.key_lv0
    osk.key_lv1[:] = sk.key_lv1
    rng = np.random.default_rng(3)
    bits = rng.integers(0, 2, 4096).astype(bool)
    cts = sk.encrypt_bool(bits, seed=10)
    assert cts.shape == (4096, 701) and cts.dtype == np.uint32
    assert np.array_equal(osk.decrypt_bool(cts[:256]), bits[:256])  # the oracle reads what the client wrote
    assert np.array_equal(sk.decrypt_bool(cts), bits)
    # noise: phase - mu is N(0, alpha) on the torus (alpha = 2e-5 -> sigma = 85,899 LSB)
    err = (sk.phase(cts) - C.f64_to_torus(np.where(bits, 0.125, -0.125))).view(np.int32).astype(np.float64)
    assert abs(err.mean()) < 6 * 85899 / np.sqrt(len(err))
    assert 0.9 < err.std() / (P.alpha_lv0 * 2**32) < 1.1
    # masks are uniform words and differ between ciphertexts and seeds
    assert len(np.unique(cts[:, 0])) > 4000
    assert not np.array_equal(cts, sk.encrypt_bool(bits, seed=11))
    assert np.array_equal(cts, sk.encrypt_bool(bits, seed=10))  # a seed fixes the ciphertexts
    msgs = rng.integers(0, 16, 512)
    P4 = R.params.SECURITY_UINT4
    sk4 = C.SecretKey.new(P4, seed=12)
    assert np.array_equal(sk4.decrypt_lwe_message(sk4.encrypt_lwe_message(msgs + 32, 16, seed=13), 16), msgs)
    with pytest.raises(ValueError):
        C.SecretKey(P, np.full(700, 2), np.zeros(1024))


def test_default_randomness_is_the_os_csprng(O, C):
    """seed=None (the default of SecretKey.new / encrypt_* / cloud_key) draws from os.urandom, as the reference's
    thread_rng is OS-seeded (tlwe.rs:38): keys and ciphertexts differ from call to call, the distributions are
    the reference's (uniform bits / words, N(0, alpha) noise on the torus)."""
    import rs_tfhe_amd as R

    P = R.params.SECURITY_128_BIT
    assert isinstance(C._rng(None), C.OsRng) and isinstance(C._rng(5), np.random.Generator)
    a, b = C.SecretKey.new(P), C.SecretKey.new(P)
    assert not np.array_equal(a.key_lv0, b.key_lv0) and not np.array_equal(a.key_lv1, b.key_lv1)
    assert set(np.unique(a.key_lv1)) == {0, 1} and 400 < int(a.key_lv1.sum()) < 624  # 6.9 sigma of Binomial(1024, 1/2)
    bits = np.random.default_rng(1).integers(0, 2, 8192).astype(bool)
    c1, c2 = a.encrypt_bool(bits), a.encrypt_bool(bits)
    assert not np.array_equal(c1, c2)
    assert np.array_equal(a.decrypt_bool(c1), bits) and np.array_equal(a.decrypt_bool(c2), bits)
    err = (a.phase(c1) - C.f64_to_torus(np.where(bits, 0.125, -0.125))).view(np.int32).astype(np.float64)
    sigma = P.alpha_lv0 * 2**32
    assert abs(err.mean()) < 6 * sigma / np.sqrt(len(err)) and 0.95 < err.std() / sigma < 1.05
    # Box-Muller tails: about 0.27 % beyond 3 sigma
    g = C.OsRng().normal(0.0, 1.0, 200001)
    assert len(g) == 200001 and abs(g.mean()) < 0.02 and 0.98 < g.std() < 1.02 and 0.0015 < (np.abs(g) > 3).mean() < 0.0045
    w = C.OsRng().integers(0, 1 << 32, (64, 700))
    assert w.shape == (64, 700) and len(np.unique(w)) > 44000 and abs(w.mean() / 2**31 - 1) < 0.02
