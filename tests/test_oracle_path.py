"""Oracle whole-path tests: the reference's randomized decrypt-equality unit tests
 restated with seeded keys, plus the golden stage vectors."""
import numpy as np
import pytest

from conftest import oracle_keys, signed_diff

N = 1024


# ---- constants and golden stage vectors ------------------------------------------
def test_constants(O, golden):
    g = golden["stage"]
    got = np.array([O.f64_to_torus(float(x)) for x in g["const_f64_to_torus_in"]], np.uint32)
    assert np.array_equal(got, g["const_f64_to_torus_out"])
    assert O.f64_to_torus(0.125) == 0x20000000 and O.f64_to_torus(-0.125) == 0xE0000000
    assert O.f64_to_torus(0.25) == 0x40000000 and O.f64_to_torus(-0.25) == 0xC0000000
    assert O.gen_decomposition_offset(3, 6) == 0x82080000
    assert O.gen_decomposition_offset(1, 22) == 0x80000000
    for (l, bg), off in zip(g["const_decomp_offset_lbg"], g["const_decomp_offset"]):
        assert O.gen_decomposition_offset(int(l), int(bg)) == int(off)
    tv = O.gen_testvec()
    assert not tv[0].any() and (tv[1] == 0x20000000).all()


def test_div_round_cases(O):
    """src/lut/generator.rs:350-356"""
    assert O.div_round(10, 3) == 3
    assert O.div_round(11, 3) == 4
    assert O.div_round(12, 3) == 4
    assert O.div_round(1, 2) == 1
    assert O.div_round(0, 5) == 0


def test_golden_rotation(O, golden):
    g = golden["stage"]
    for k, exp in zip(g["rot_k"], g["rot_out"]):
        assert np.array_equal(O.poly_mul_with_x_k(g["rot_in"], int(k)), exp)
    p = g["rot_in"]
    # quirk Q1/Q2: k=0 and k=2N are the identity, k=N is MAX - x (not -x)
    assert np.array_equal(O.poly_mul_with_x_k(p, 0), p)
    assert np.array_equal(O.poly_mul_with_x_k(p, 2 * N), p)
    assert np.array_equal(O.poly_mul_with_x_k(p, N), ~p)


def test_golden_decomposition_and_recomposition(O, golden):
    g = golden["stage"]
    for l, bg in ((3, 6), (2, 10), (1, 22)):
        off = O.gen_decomposition_offset(l, bg)
        dec = O.decomposition(g["dec_in"], l, bg, off)
        assert np.array_equal(dec, g[f"dec_out_{l}_{bg}"])
        # digits are signed in [-Bg/2, Bg/2) and recompose to the input up to 2^(32-l*bg)
        d = dec.view(np.int32).astype(np.int64)
        assert d.min() >= -(1 << (bg - 1)) and d.max() < (1 << (bg - 1))
        for half in range(2):
            rec = np.zeros(N, np.int64)
            for i in range(l):
                rec += d[half * l + i] << (32 - (i + 1) * bg)
            err = (rec.astype(np.uint64).astype(np.uint32) - g["dec_in"][half]).astype(np.int32)
            assert np.abs(err.astype(np.int64)).max() <= (1 << (32 - l * bg)) if l * bg < 32 else True


def test_golden_sample_extract(O, golden):
    g = golden["stage"]
    for k, exp in zip(g["se_k"], g["se_out"]):
        assert np.array_equal(O.sample_extract_index(g["se_in"], int(k)), exp)
    for k, exp in zip((0, 3, 699), g["se2_out_n700"]):
        assert np.array_equal(O.sample_extract_index_2(g["se_in"], k, 700), exp)
    t = g["se_in"]
    e = O.sample_extract_index(t, 0)
    assert e[0] == t[0, 0] and e[N] == t[1, 0]
    assert np.array_equal(e[1:N], ~t[0, :0:-1])  # MAX - a[N-i]


def test_golden_gate_prep(O, golden):
    g = golden["stage"]
    for op in range(10):
        assert np.array_equal(O.gate_prep(op, g["prep_a"], g["prep_b"], 16), g["prep_out"][op])
    a, b = g["prep_a"], g["prep_b"]
    nand = O.gate_prep(O.GATE_NAND, a, b, 16)
    exp = (0 - (a.astype(np.int64) + b.astype(np.int64))).astype(np.uint64).astype(np.uint32)
    exp[16] += np.uint32(0x20000000)
    assert np.array_equal(nand, exp)
    with pytest.raises(ValueError):
        O.gate_prep(99, a, b, 16)


def test_golden_luts(O, golden):
    g = golden["stage"]
    assert np.array_equal(O.lut_generate(lambda x: x, 2), g["lut_id_m2"])
    assert np.array_equal(O.lut_generate(lambda x: 1 - x, 2), g["lut_not_m2"])
    assert np.array_equal(O.lut_generate(lambda x: x, 4), g["lut_id_m4"])
    assert np.array_equal(O.lut_generate(lambda x: (x * x) % 16, 16), g["lut_sq_m16"])
    assert np.array_equal(O.lut_generate(lambda x: x, 3), g["lut_id_m3"])
    # structure for m=16: 64-wide steps, offset 32, tail negated
    t = g["lut_sq_m16"][1]
    assert t[0] == O.lut_encode(0, 16) and t[32] == O.lut_encode(1, 16)
    assert t[N - 1] == (0 - O.lut_encode(0, 16)) & 0xFFFFFFFF


def test_encoder_roundtrips(O):
    """src/lut/encoder.rs:124-160"""
    for m in (2, 4, 16):
        for x in range(m):
            assert O.lut_decode(O.lut_encode(x, m), m) == x


def test_golden_toy_bootstrap(O, golden):
    """The committed n=4 instance reproduces bit-for-bit from its own stored key."""
    g = golden["toy"]
    n, l, bgbit, basebit, t = (int(v) for v in g["params"])
    P = O.Params("TOY_N4", n, l, bgbit, basebit, t, 2.0e-5, 2.0e-8)

    class CK:
        pass

    ck = O.CloudKey.__new__(O.CloudKey)
    ck.params = P
    ck.decomposition_offset = int(g["offset"][0])
    ck.blind_rotate_testvec = np.ascontiguousarray(g["testvec"])
    ck.bootstrapping_key = np.ascontiguousarray(g["bsk"])
    ck.bootstrapping_key_time = np.ascontiguousarray(g["bsk_time"])
    ck.key_switching_key = np.ascontiguousarray(g["ksk"])
    assert np.array_equal(O.batch_blind_rotate(ck, g["cts"]), g["blind_rotate"])
    assert np.array_equal(g["blind_rotate"], g["blind_rotate_exact"])  # f64 path == exact integers
    assert np.array_equal(O.batch_bootstrap(ck, g["cts"]), g["bootstrap"])
    assert np.array_equal(O.batch_bootstrap(ck, g["cts"], keyswitch=False), g["bootstrap_noks"])
    for op in range(10):
        assert np.array_equal(O.batch_gate(ck, op, g["cts"], g["cts2"]), g[f"gate_{op}"])
    assert np.array_equal(O.batch_bootstrap(ck, g["cts"], testvec=g["lut"]), g["bootstrap_lut"])
    assert np.array_equal(O.batch_mux(ck, g["cts"], g["cts2"], g["cts"][::-1].copy(), naive=True), g["mux_naive"])


# ---- property tests mirroring the reference's unit tests --------------------------
def test_external_product_fft_is_exact(O, keys128):
    """bgbit=6,l=3: the f64 FFT product equals the exact integer product (0 LSB)."""
    sk, ck = keys128
    P = ck.params
    rng = np.random.default_rng(7)
    for i in (0, 1, 350, 699):
        t = rng.integers(0, 2**32, (2, N), dtype=np.uint64).astype(np.uint32)
        a = O.external_product_fft(ck.bootstrapping_key[i], t, P.l, P.bgbit, ck.decomposition_offset)
        b = O.external_product_exact(ck.bootstrapping_key_time[i], t, P.l, P.bgbit, ck.decomposition_offset)
        assert signed_diff(a, b) == 0


def test_external_product_preserves_plaintext(O, keys128):
    """trgsw.rs:427-466: TRGSW(1) (x) TRLWE(m) decrypts to m; TRGSW(0) to ~0."""
    sk, ck = keys128
    P = ck.params
    rng = np.random.default_rng(8)
    ones = [i for i in range(P.n) if sk.key_lv0[i] == 1][:2]
    zeros = [i for i in range(P.n) if sk.key_lv0[i] == 0][:2]
    msg = np.where(rng.integers(0, 2, N).astype(bool), np.uint32(0x20000000), np.uint32(0xE0000000))
    a = rng.integers(0, 2**32, N, dtype=np.uint64).astype(np.uint32)
    trlwe = np.stack([a, (O.negacyclic_schoolbook(a, sk.key_lv1) + msg).astype(np.uint32)])
    for i in ones:
        out = O.external_product_fft(ck.bootstrapping_key[i], trlwe, P.l, P.bgbit, ck.decomposition_offset)
        ph = sk.trlwe_phase(out)
        assert np.array_equal(ph.view(np.int32) >= 0, msg.view(np.int32) >= 0)
    for i in zeros:
        out = O.external_product_fft(ck.bootstrapping_key[i], trlwe, P.l, P.bgbit, ck.decomposition_offset)
        assert np.abs(sk.trlwe_phase(out).view(np.int32).astype(np.int64)).max() < (1 << 27)


def test_cmux_selects(O, keys128):
    """trgsw.rs:469-505"""
    sk, ck = keys128
    P = ck.params
    rng = np.random.default_rng(9)

    def enc(bits):
        a = rng.integers(0, 2**32, N, dtype=np.uint64).astype(np.uint32)
        m = np.where(bits, np.uint32(0x20000000), np.uint32(0xE0000000))
        return np.stack([a, (O.negacyclic_schoolbook(a, sk.key_lv1) + m).astype(np.uint32)])

    b1 = rng.integers(0, 2, N).astype(bool)
    b2 = rng.integers(0, 2, N).astype(bool)
    c1, c2 = enc(b1), enc(b2)
    for i in range(4):
        out = O.cmux(c1, c2, ck.bootstrapping_key[i], P.l, P.bgbit, ck.decomposition_offset)
        got = sk.trlwe_phase(out).view(np.int32) >= 0
        assert np.array_equal(got, b2 if sk.key_lv0[i] else b1)


def test_blind_rotate_and_extract_decrypts(O, keys128):
    """trgsw.rs:508-529 + fft path == exact path for the whole 700-step chain."""
    sk, ck = keys128
    bits = np.array([True, False, True])
    cts = sk.encrypt_bool(bits, 11)
    for bit, ct in zip(bits, cts):
        trlwe = O.blind_rotate(ck, ct)
        assert sk.decrypt_bool_lv1(O.sample_extract_index(trlwe, 0)) == bit
    assert np.array_equal(O.blind_rotate(ck, cts[0]), O.blind_rotate(ck, cts[0], exact=True))


def test_identity_key_switching_decrypts(O, keys128):
    """trgsw.rs:532-546"""
    sk, ck = keys128
    rng = np.random.default_rng(12)
    for bit in (True, False, True, False):
        a = rng.integers(0, 2**32, N, dtype=np.uint64).astype(np.uint32)
        mu = np.uint32(0x20000000 if bit else 0xE0000000)
        b = np.uint32((a.astype(np.uint64) * sk.key_lv1).sum() & 0xFFFFFFFF) + mu
        lv1 = np.concatenate([a, [b]]).astype(np.uint32)
        assert sk.decrypt_bool_lv1(lv1) == bit
        out = O.identity_key_switching(ck, lv1)
        assert sk.decrypt_bool(out)[0] == bit


@pytest.mark.parametrize("op", range(10))
def test_gate_truth_tables(O, keys128, op):
    """gates.rs:559-653 + the four gates the reference leaves untested in batch form."""
    sk, ck = keys128
    A = np.array([0, 0, 1, 1], bool)
    B = np.array([0, 1, 0, 1], bool)
    out = O.batch_gate(ck, op, sk.encrypt_bool(A, 100 + op), sk.encrypt_bool(B, 200 + op))
    exp = np.array([O.GATE_TRUTH[op](bool(a), bool(b)) for a, b in zip(A, B)])
    assert np.array_equal(sk.decrypt_bool(out), exp)


def test_mux_naive_truth_table(O, keys128):
    """gates.rs:656-681; Gates::mux is the reference formula (quirk Q5) -> determinism only."""
    sk, ck = keys128
    A = np.array([0, 0, 0, 0, 1, 1, 1, 1], bool)
    B = np.array([0, 0, 1, 1, 0, 0, 1, 1], bool)
    Cc = np.array([0, 1, 0, 1, 0, 1, 0, 1], bool)
    ca, cb, cc = sk.encrypt_bool(A, 31), sk.encrypt_bool(B, 32), sk.encrypt_bool(Cc, 33)
    out = O.batch_mux(ck, ca, cb, cc, naive=True)
    assert np.array_equal(sk.decrypt_bool(out), np.where(A, B, Cc))
    m1 = O.batch_mux(ck, ca[:2], cb[:2], cc[:2], naive=False)
    m2 = O.batch_mux(ck, ca[:2], cb[:2], cc[:2], naive=False)
    assert np.array_equal(m1, m2)


def test_lut_bootstrap_binary(O, keys128):
    """bootstrap/lut.rs:142-254: identity / NOT / constant with message_modulus = 2."""
    sk, ck = keys128
    for f, name in ((lambda x: x, "id"), (lambda x: 1 - x, "not"), (lambda x: 1, "const")):
        lut = O.lut_generate(f, 2)
        msgs = np.array([0, 1, 1, 0])
        cts = sk.encrypt_lwe_message(msgs, 2, 50)
        out = O.batch_bootstrap(ck, cts, testvec=lut)
        assert np.array_equal(sk.decrypt_lwe_message(out, 2), np.array([f(int(m)) % 2 for m in msgs])), name


def test_xor_80bit(O, keys80):
    sk, ck = keys80
    A = np.array([0, 0, 1, 1], bool)
    B = np.array([0, 1, 0, 1], bool)
    out = O.batch_gate(ck, O.GATE_XOR, sk.encrypt_bool(A, 1), sk.encrypt_bool(B, 2))
    assert np.array_equal(sk.decrypt_bool(out), A ^ B)


def test_pbs_uint4(O, keys_uint4):
    """BASELINE config 4 semantics: m=16, f(x)=x^2 mod 16 with the SECURITY_UINT4 numbers."""
    sk, ck = keys_uint4
    P = ck.params
    msgs = np.arange(16)
    cts = sk.encrypt_lwe_message(msgs, 16, 60)
    assert np.array_equal(sk.decrypt_lwe_message(cts, 16), msgs)
    lut = O.lut_generate(lambda x: (x * x) % 16, 16)
    out = O.batch_bootstrap(ck, cts, testvec=lut)
    assert np.array_equal(sk.decrypt_lwe_message(out, 16), (msgs * msgs) % 16)
    # f64 external product error at bgbit=22 is bounded (~2^7 LSB, allow 2^9)
    rng = np.random.default_rng(13)
    t = rng.integers(0, 2**32, (2, N), dtype=np.uint64).astype(np.uint32)
    a = O.external_product_fft(ck.bootstrapping_key[3], t, P.l, P.bgbit, ck.decomposition_offset)
    b = O.external_product_exact(ck.bootstrapping_key_time[3], t, P.l, P.bgbit, ck.decomposition_offset)
    assert signed_diff(a, b) <= 512
