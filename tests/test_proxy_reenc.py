"""Proxy re-encryption (the reference's feature `proxy-reenc`, src/proxy_reenc.rs): the oracle's restatement against
the properties the reference's own tests assert (proxy_reenc.rs:520-703), the product's client-side key generation
checked through the oracle on CPU, and -- on the GPU -- `tfhe_hip_batch_reencrypt` word for word against
reencrypt_tlwe_lv0 (proxy_reenc.rs:468-510) on every key-switch kernel."""
import numpy as np
import pytest


def _product_params(op):
    import rs_tfhe_amd as R

    return R.params.PARAM_SETS[op.name]


def _product_sk(sk):
    import rs_tfhe_amd as R

    return R.SecretKey(_product_params(sk.params), sk.key_lv0, sk.key_lv1)


@pytest.fixture(scope="module")
def parties(O):
    P = O.SECURITY_128_BIT
    return P, O.SecretKey(P, 101), O.SecretKey(P, 202), O.SecretKey(P, 303)


# ---- oracle (CPU) ------------------------------------------------------------------------------------------------
def test_oracle_symmetric_reencryption_decrypts_under_the_target_key(O, parties):
    """test_proxy_reencryption_symmetric (proxy_reenc.rs:584-604) + test_proxy_reencryption_key_generation (:639-653)."""
    P, alice, bob, _ = parties
    key = O.gen_reenc_key(P, alice.key_lv0, 7, key_to=bob.key_lv0)
    assert key.shape == (P.base * P.t * P.n, P.n + 1)
    assert not key.reshape(P.n, P.t, P.base, P.n + 1)[:, :, 0, :].any()  # the k = 0 entries are never written (:311-313)
    bits = np.random.default_rng(1).integers(0, 2, 96).astype(bool)
    ca = alice.encrypt_bool(bits, 5)
    cb = O.reencrypt_tlwe_lv0(P, key, ca)
    assert np.array_equal(alice.decrypt_bool(ca), bits)
    assert np.array_equal(bob.decrypt_bool(cb), bits)
    assert not np.array_equal(alice.decrypt_bool(cb), bits)  # and no longer under the source key


def test_oracle_public_key_encryption_and_asymmetric_reencryption(O, parties):
    """test_public_key_encryption / _multiple (:527-556), test_proxy_reencryption_asymmetric[_multiple] (:558-636: the
    reference accepts > 90 % there -- the public-key noise is the sum of ~n fresh encryptions)."""
    P, alice, bob, _ = parties
    pk = O.PublicKeyLv0(bob, 9)
    assert pk.encryptions.shape == (2 * P.n, P.n + 1)
    bits = np.random.default_rng(2).integers(0, 2, 200).astype(bool)
    assert (bob.decrypt_bool(pk.encrypt_bool(bits, P.alpha_lv0, 11)) == bits).mean() > 0.90
    key = O.gen_reenc_key(P, alice.key_lv0, 13, public_key_to=pk)
    cb = O.reencrypt_tlwe_lv0(P, key, alice.encrypt_bool(bits, 15))
    assert (bob.decrypt_bool(cb) == bits).mean() > 0.90


def test_oracle_reencryption_chain(O, parties):
    """test_proxy_reencryption_chain_asymmetric (:658-680) in the symmetric mode (exact margins): Alice -> Bob -> Carol."""
    P, alice, bob, carol = parties
    bits = np.random.default_rng(3).integers(0, 2, 64).astype(bool)
    ab = O.gen_reenc_key(P, alice.key_lv0, 21, key_to=bob.key_lv0)
    bc = O.gen_reenc_key(P, bob.key_lv0, 22, key_to=carol.key_lv0)
    cc = O.reencrypt_tlwe_lv0(P, bc, O.reencrypt_tlwe_lv0(P, ab, alice.encrypt_bool(bits, 23)))
    assert np.array_equal(carol.decrypt_bool(cc), bits)


def test_oracle_reencryption_is_the_key_switch_of_a_zero_padded_sample(O, parties):
    """What the GPU path relies on (include/tfhe_hip.h): reencrypt_tlwe_lv0 == identity_key_switching (trgsw.rs:332-360)
    of the sample padded to N coefficients with zeros, under a key-switching key whose rows i >= n are unused."""
    P, alice, bob, _ = parties
    N = 1024
    key = O.gen_reenc_key(P, alice.key_lv0, 31, key_to=bob.key_lv0)
    ksk = np.zeros((N * P.t * P.base, P.n + 1), np.uint32)
    ksk[: len(key)] = key
    ksk[len(key):] = 0xDEADBEEF  # never selected
    ca = np.random.default_rng(4).integers(0, 2**32, (40, P.n + 1), dtype=np.uint64).astype(np.uint32)
    lv1 = np.zeros((40, N + 1), np.uint32)
    lv1[:, : P.n] = ca[:, : P.n]
    lv1[:, N] = ca[:, P.n]

    class _CK:  # the two fields identity_key_switching reads
        params, key_switching_key = P, ksk

    assert np.array_equal(O.batch_identity_key_switching(_CK, lv1), O.reencrypt_tlwe_lv0(P, key, ca))


# ---- product client side (numpy key generation), through the oracle's re-encryption ----------------------------------
def test_product_key_generation_symmetric_and_asymmetric(O, parties):
    from rs_tfhe_amd import proxy_reenc as PR

    P, alice, bob, _ = parties
    pa, pb = _product_sk(alice), _product_sk(bob)
    bits = np.random.default_rng(5).integers(0, 2, 128).astype(bool)
    ca = alice.encrypt_bool(bits, 41)
    ks = PR.ProxyReencryptionKey.new_symmetric(pa, pb, seed=42)
    assert (ks.base, ks.t) == (P.base, P.t) and ks.key_encryptions.shape == (P.base * P.t * P.n, P.n + 1)
    assert not ks.key_encryptions.reshape(P.n, P.t, P.base, P.n + 1)[:, :, 0, :].any()
    assert np.array_equal(bob.decrypt_bool(O.reencrypt_tlwe_lv0(P, ks.key_encryptions, ca)), bits)
    # every entry decrypts to its plaintext k * s_i / base^(j+1) within the key-switch noise (:316, :414)
    e = ks.key_encryptions.reshape(P.n, P.t, P.base, P.n + 1)
    i, j, k = 17, 2, 3
    want = (k * int(alice.key_lv0[i])) / float(1 << ((j + 1) * P.basebit))
    got = bob.phase(e[i, j, k][None])[0].astype(np.int32) / 2.0**32
    assert abs(got - want) < 8 * P.alpha_lv0
    # public key: encryptions of zero, public-key encryption, asymmetric key (> 90 %: the reference's own bar)
    pk = PR.PublicKeyLv0.new(pb, seed=43)
    assert pk.encryptions.shape == (2 * P.n, P.n + 1)
    assert np.abs(bob.phase(pk.encryptions).view(np.int32) / 2.0**32).max() < 8 * P.alpha_lv0
    assert (bob.decrypt_bool(pk.encrypt_bool(bits, P.alpha_lv0, seed=44)) == bits).mean() > 0.90
    ka = PR.ProxyReencryptionKey.new_asymmetric(pa, pk, seed=45)
    assert (bob.decrypt_bool(O.reencrypt_tlwe_lv0(P, ka.key_encryptions, ca)) == bits).mean() > 0.90
    # the combination draws: each encryption of zero joins with probability 1/2, either sign with probability 1/2
    c = pk.encrypt_f64(np.zeros(4), 0.0, seed=46)
    assert len({row.tobytes() for row in c}) == 4  # distinct combinations


# ---- GPU -------------------------------------------------------------------------------------------------------------
def _edge(rng, count, n):
    a = rng.integers(0, 2**32, (count, n + 1), dtype=np.uint64).astype(np.uint32)
    a[0, :n] = 0
    a[-1, :n] = 0xFFFFFFFF
    return a


@pytest.mark.gpu
@pytest.mark.parametrize("setname,kernel", [
    ("SECURITY_128_BIT", "auto"),     # split below 64, matrix cores with K chunks above
    ("SECURITY_128_BIT", "mfma"),
    ("SECURITY_128_BIT", "b4"),
    ("SECURITY_128_BIT", "generic"),
    ("SECURITY_128_BIT", "split"),
    ("SECURITY_UINT4", "auto"),       # base 32: split, then the column-sliced kernel
    ("SECURITY_UINT4", "sliced"),
    ("SECURITY_UINT2", "sliced"),     # base 16
    ("SECURITY_UINT3", "sliced"),     # base 64 (ring of two pairs), t = 2
])
def test_gpu_reencrypt_bit_exact(O, monkeypatch, setname, kernel):
    """tfhe_hip_batch_reencrypt == reencrypt_tlwe_lv0 (proxy_reenc.rs:468-510) word for word, on every key-switch
    kernel (forced by TFHE_HIP_KS_KERNEL), at ragged counts; the outputs decrypt under the target key."""
    import rs_tfhe_amd as R

    op = getattr(O, setname)
    alice, bob = O.SecretKey(op, 501), O.SecretKey(op, 502)
    key = O.gen_reenc_key(op, alice.key_lv0, 503, key_to=bob.key_lv0)
    monkeypatch.setenv("TFHE_HIP_KS_KERNEL", kernel)
    eng = R.Engine(_product_params(op), 0)
    assert not eng.reenc_key_is_loaded()
    eng.load_reenc_key(key)
    assert eng.reenc_key_is_loaded()
    rng = np.random.default_rng(77)
    for count in (1, 63, 64, 385, 1301):
        ca = _edge(rng, count, op.n)
        assert np.array_equal(eng.batch_reencrypt(ca), O.reencrypt_tlwe_lv0(op, key, ca)), (setname, kernel, count)
    bits = rng.integers(0, 2, 500).astype(bool)
    cb = eng.batch_reencrypt(alice.encrypt_bool(bits, 504))
    assert np.array_equal(bob.decrypt_bool(cb), bits)
    eng.close()


@pytest.mark.gpu
def test_gpu_reencrypt_handles_and_errors(O, keys128):
    """A handle holds EITHER a cloud key OR a re-encryption key (include/tfhe_hip.h): the wrong family of calls returns
    TFHE_HIP_ENOKEY; key views keep both kinds resident on one context; the device-resident form only enqueues."""
    import torch

    import rs_tfhe_amd as R
    from rs_tfhe_amd import _capi
    from tests.test_gpu_parity import _cloud_key

    sk, ck = keys128
    P = O.SECURITY_128_BIT
    bob = O.SecretKey(P, 602)
    key = O.gen_reenc_key(P, sk.key_lv0, 603, key_to=bob.key_lv0)
    pk = _cloud_key(ck)
    eng = R.Engine(pk.params, 0)
    eng.load_cloud_key(pk)
    bits = np.random.default_rng(6).integers(0, 2, 300).astype(bool)
    ca = sk.encrypt_bool(bits, 604)
    with pytest.raises(_capi.TfheHipError, match="re-encryption key not loaded"):
        eng.batch_reencrypt(ca)
    view = eng.new_key_view()  # the re-encryption key beside the cloud key, same context
    view.load_reenc_key(key)
    with pytest.raises(_capi.TfheHipError, match="cloud key not loaded"):
        view.batch_gate(R.engine.NAND, ca, ca)
    # a gate under Alice's cloud key, then the result handed to Bob: NAND(x, x) = NOT x
    g = eng.batch_gate(R.engine.NAND, ca, ca)
    out = view.batch_reencrypt(g)
    assert np.array_equal(out, O.reencrypt_tlwe_lv0(P, key, g))
    assert np.array_equal(bob.decrypt_bool(out), ~bits)
    # device-resident, on a caller's stream
    ta = torch.from_numpy(g.view(np.int32)).cuda()
    to = torch.empty_like(ta)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        view.batch_reencrypt_dev(ta, to, s)
    s.synchronize()
    assert np.array_equal(to.cpu().numpy().view(np.uint32), out)
    # sets whose n exceeds N = 1024 (the key switch's source rows) are refused with a message
    big = R.Engine(R.params.SECURITY_UINT7, 0)
    with pytest.raises(_capi.TfheHipError, match="n <= N"):
        big.load_reenc_key(np.zeros((1160 * 3 * 128, 1161), np.uint32))
    big.close()
    # loading a cloud key into the view drops the re-encryption key
    view.load_cloud_key(pk)
    assert not view.reenc_key_is_loaded()
    with pytest.raises(_capi.TfheHipError, match="re-encryption key not loaded"):
        view.batch_reencrypt(ca)
    assert np.array_equal(view.batch_gate(R.engine.NAND, ca, ca), g)
    view.close()
    eng.close()


@pytest.mark.gpu
def test_gpu_reencrypt_reference_flow_and_custom_params(O):
    """The reference's README flow with the product alone (proxy_reenc.rs:23-66, test_custom_params :683-701): keys,
    encryption, asymmetric and symmetric re-encryption keys, a chain, and a key with its own (basebit, t)."""
    import rs_tfhe_amd as R
    from rs_tfhe_amd import proxy_reenc as PR

    P = R.params.SECURITY_128_BIT
    alice, bob, carol = (R.SecretKey.new(P, seed=s) for s in (701, 702, 703))
    bits = np.random.default_rng(8).integers(0, 2, 400).astype(bool)
    ca = alice.encrypt_bool(bits, 704)
    ab = PR.ProxyReencryptionKey.new_symmetric(alice, bob, seed=705)
    bc = PR.ProxyReencryptionKey.new_symmetric(bob, carol, seed=706)
    cb = PR.reencrypt_tlwe_lv0(ca, ab)
    assert np.array_equal(bob.decrypt_bool(cb), bits)
    assert np.array_equal(carol.decrypt_bool(bc.reencrypt(cb)), bits)
    one = PR.reencrypt_tlwe_lv0(ca[0], ab)  # a single ciphertext, as the reference's signature takes it
    assert one.shape == (P.n + 1,) and np.array_equal(one, cb[0])
    pub = PR.PublicKeyLv0.new(bob, seed=707)
    asym = PR.ProxyReencryptionKey.new_asymmetric(alice, pub, seed=708)
    assert (bob.decrypt_bool(asym.reencrypt(ca)) == bits).mean() > 0.90  # the reference's own bar (:627-632)
    custom = PR.ProxyReencryptionKey.new_symmetric_with_params(alice, bob, P.alpha_lv0 * 0.8, 3, 6, seed=709)
    assert (custom.base, custom.t) == (8, 6)
    cc = custom.reencrypt(ca)
    assert np.array_equal(bob.decrypt_bool(cc), bits)
    op = O.Params("custom", P.n, P.l, P.bgbit, 3, 6, P.alpha_lv0, P.alpha_lv1) if hasattr(O, "Params") else None
    if op is not None:
        assert np.array_equal(cc, O.reencrypt_tlwe_lv0(op, custom.key_encryptions, ca))
    for k in (ab, bc, asym, custom):
        k.close()
