"""Oracle FFT layer vs the reference's own FFT test inputs and tolerances
(src/fft/mod.rs:119-238, src/fft/klemsa.rs:183-202, src/fft/processors.rs:783-855),
vs the DFT definition, and vs the reference's SPQLIOS C++/asm build (oracle/_ref)."""
import numpy as np
import pytest

from conftest import signed_diff

N = 1024


def test_dft_definition(O):
    """orc_cfft512 computes the un-normalised DFT rustfft computes (klemsa.rs:62-65)."""
    rng = np.random.default_rng(1)
    x = rng.standard_normal(512) + 1j * rng.standard_normal(512)
    assert np.abs(O.cfft512(x) - np.fft.fft(x)).max() < 1e-11
    assert np.abs(O.cfft512(x, inverse=True) - np.fft.ifft(x) * 512).max() < 1e-11


def test_klemsa_spectrum_layout(O):
    """bin k = evaluation at exp(i*pi*(1-4k)/N) * 2, stored re[0..512] || im[0..512]."""
    rng = np.random.default_rng(2)
    p = rng.integers(0, 2**32, N, dtype=np.uint64).astype(np.uint32)
    spec = O.klemsa_ifft(p)
    x = p.view(np.int32).astype(np.float64)
    z = (x[:512] + 1j * x[512:]) * np.exp(1j * np.pi * np.arange(512) / N)
    ref = 2.0 * np.fft.fft(z)
    scale = np.abs(ref).max()
    assert np.abs(spec[:512] - ref.real).max() < 1e-12 * scale
    assert np.abs(spec[512:] - ref.imag).max() < 1e-12 * scale


def test_kat_klemsa_roundtrip(O, golden):
    """klemsa.rs:183-202: in[0]=2^31, in[5]=2^30, |out-in| < 2."""
    x = golden["stage"]["kat_klemsa_roundtrip"]
    assert signed_diff(O.klemsa_fft(O.klemsa_ifft(x)), x) < 2


def test_kat_delta(O, golden):
    """fft/mod.rs:162-177: delta 1000, |diff| < 10."""
    x = golden["stage"]["kat_delta"]
    assert signed_diff(O.klemsa_fft(O.klemsa_ifft(x)), x) < 10


def test_random_roundtrip(O):
    """fft/mod.rs:119-133, 180-210: random torus, |diff| < 2."""
    rng = np.random.default_rng(3)
    for _ in range(20):
        a = rng.integers(0, 2**32, N, dtype=np.uint64).astype(np.uint32)
        assert signed_diff(O.klemsa_fft(O.klemsa_ifft(a)), a) < 2


@pytest.mark.parametrize("pair", ["consistency", "dense"])
def test_kat_poly_mul_vs_schoolbook(O, golden, pair):
    """processors.rs:809-813, 850-855 deterministic inputs (< 2^20 each: sums < 2^50, exact
    in f64), expected = exact schoolbook, |diff| < 2."""
    g = golden["stage"]
    a, b = g[f"kat_{pair}_a"], g[f"kat_{pair}_b"]
    exp = g[f"kat_{pair}_expected"]
    assert np.array_equal(O.negacyclic_schoolbook(a, b), exp)
    assert signed_diff(O.klemsa_poly_mul(a, b), exp) < 2


def test_kat_sparse_is_integer_only(O, golden):
    """processors.rs:783-786 (orphan, never compiled): a=2^28, b=2^27 every 20th slot.
    Every product is 2^55 = 0 mod 2^32 and the sums reach 2^60 > 2^53, so no f64 FFT can
    be held to +-1 here; the vector pins the exact schoolbook only (expected: all zero)."""
    g = golden["stage"]
    exp = O.negacyclic_schoolbook(g["kat_sparse_a"], g["kat_sparse_b"])
    assert np.array_equal(exp, g["kat_sparse_expected"]) and not exp.any()


def test_random_poly_mul_vs_schoolbook(O):
    """fft/mod.rs:136-159, 213-238: a uniform, b < Bg; tolerance < 2 (measured 0)."""
    rng = np.random.default_rng(4)
    worst = 0
    for _ in range(30):
        a = rng.integers(0, 2**32, N, dtype=np.uint64).astype(np.uint32)
        b = rng.integers(0, 64, N, dtype=np.uint64).astype(np.uint32)
        worst = max(worst, signed_diff(O.klemsa_poly_mul(a, b), O.negacyclic_schoolbook(a, b)))
    assert worst < 2


def test_schoolbook_independent(O):
    """the oracle's schoolbook equals an independent numpy big-int negacyclic product mod 2^32."""
    rng = np.random.default_rng(5)
    a = rng.integers(0, 2**32, N, dtype=np.uint64).astype(np.uint32)
    b = rng.integers(0, 2**32, N, dtype=np.uint64).astype(np.uint32)
    full = np.zeros(2 * N, dtype=object)
    A = a.astype(object)
    for i in range(N):
        full[i:i + N] += A * int(b[i])
    ref = np.array([(int(full[i]) - int(full[i + N])) % (1 << 32) for i in range(N)], dtype=np.uint64).astype(np.uint32)
    assert np.array_equal(O.negacyclic_schoolbook(a, b), ref)


def test_reference_spqlios_cross_check(O, golden):
    """The reference's own SPQLIOS negacyclic FFT (compiled from /root/reference into
    oracle/_ref) agrees with the oracle's products within its truncation (+-1 LSB)."""
    if not O.ref_available():
        pytest.skip("oracle/_ref not built (reference sources absent)")
    rng = np.random.default_rng(6)
    g = golden["stage"]
    cases = [(g[f"kat_{p}_a"], g[f"kat_{p}_b"]) for p in ("consistency", "dense")]
    for _ in range(10):
        cases.append(
            (rng.integers(0, 2**32, N, dtype=np.uint64).astype(np.uint32),
             rng.integers(0, 64, N, dtype=np.uint64).astype(np.uint32))
        )
    for a, b in cases:
        ref = O.ref_poly_mul(a, b)
        assert signed_diff(ref, O.negacyclic_schoolbook(a, b)) <= 1
        assert signed_diff(ref, O.klemsa_poly_mul(a, b)) <= 1
    x = g["kat_klemsa_roundtrip"]
    assert signed_diff(O.ref_roundtrip(x), x) <= 1


def test_reference_spqlios_external_product(O, keys128):
    """A whole external product (trgsw.rs:77-116) assembled from the REFERENCE's own compiled code:
    2l calls of Spqlios_poly_mul_1024 (spqlios-wrapper.cpp:33-40) per output polynomial, on the digits
    orc_decomposition produces.  Holds orc_external_product_exact / _fft to real reference arithmetic
    for the things a second restatement could share a slip on: the sign convention of the digits
    (quirk Q3: stored as wrapped u32, read back as signed), which rows multiply which half
    (rows 0..l-1 <- a, l..2l-1 <- b, trgsw.rs:99-106) and the a / b column order.
    Tolerance: SPQLIOS truncates each product (fft_processor_spqlios.cpp:128-129): <= 1 LSB per
    product, 2l products per coefficient."""
    if not O.ref_available():
        pytest.skip("oracle/_ref not built (reference sources absent)")
    sk, ck = keys128
    P = ck.params
    rng = np.random.default_rng(21)
    for i in (0, 7, 699):
        t = rng.integers(0, 2**32, (2, N), dtype=np.uint64).astype(np.uint32)
        dec = O.decomposition(t, P.l, P.bgbit, ck.decomposition_offset)
        # digits are signed in [-Bg/2, Bg/2) once read as i32 (Q3)
        d = dec.view(np.int32)
        assert d.min() >= -(1 << (P.bgbit - 1)) and d.max() < (1 << (P.bgbit - 1))
        out = np.zeros((2, N), np.uint32)
        for r in range(2 * P.l):
            for c in range(2):
                out[c] += O.ref_poly_mul(ck.bootstrapping_key_time[i][r][c], dec[r])
        exact = O.external_product_exact(ck.bootstrapping_key_time[i], t, P.l, P.bgbit, ck.decomposition_offset)
        fft = O.external_product_fft(ck.bootstrapping_key[i], t, P.l, P.bgbit, ck.decomposition_offset)
        assert signed_diff(out, exact) <= 2 * P.l
        assert signed_diff(out, fft) <= 2 * P.l
        # and the assembly is sensitive to what it is meant to pin: swapping the halves' rows or
        # reading the digits unsigned lands far away
        swapped = np.zeros((2, N), np.uint32)
        for r in range(2 * P.l):
            for c in range(2):
                swapped[c] += O.ref_poly_mul(ck.bootstrapping_key_time[i][(r + P.l) % (2 * P.l)][c], dec[r])
        assert signed_diff(swapped, exact) > (1 << 20)


def test_reference_spqlios_pins_rotation_and_cmux(O, keys128):
    """Direction of rotation, the `Torus::MAX - x` quirk (Q1) and one chained CMUX step held to REAL reference
    arithmetic instead of a second restatement.

    poly_mul_with_x_k (trgsw.rs:307-330) claims to be multiplication by X^k in Z[X]/(X^N+1), k in [0, 2N].
    The reference's compiled SPQLIOS product with the monomial X^k mod (X^N+1) (= +X^k for k < N, -X^(k-N) for
    N <= k < 2N, +1 at k = 2N) gives the true product; the quirk makes every WRAPPED coefficient
    MAX - a = -a - 1 instead of -a.  So: equal on unwrapped coefficients, off by exactly one on wrapped ones --
    which pins which side wraps (direction) and the off-by-one itself.  SPQLIOS truncates (+-1 LSB per product,
    fft_processor_spqlios.cpp:128-129), so `a` holds multiples of 256: the reference product, rounded to the nearest
    multiple of 256, is then the exact product, and an off-by-one stands out unambiguously."""
    if not O.ref_available():
        pytest.skip("oracle/_ref not built (reference sources absent)")
    rng = np.random.default_rng(31)
    a = (rng.integers(0, 2**20, N, dtype=np.uint64) * 256).astype(np.uint32)
    for k in (0, 1, 2, N - 1, N, N + 1, 2 * N - 1, 2 * N):
        mono = np.zeros(N, np.uint32)
        kk = k % (2 * N)
        if kk < N:
            mono[kk] = 1
        else:
            mono[kk - N] = 0xFFFFFFFF  # -X^(k-N)
        raw = O.ref_poly_mul(a, mono)
        true = ((raw.astype(np.uint64) + 128) // 256 * 256).astype(np.uint32)  # exact product (wraps mod 2^32)
        assert signed_diff(raw, true) <= 1
        got = O.poly_mul_with_x_k(a, k)
        # wrapped coefficients: output index j came from a[j - k + N] negated (k < N), or the complement region (k >= N)
        j = np.arange(N)
        wrapped = (j < k) if k < N else (j >= k - N)
        if k == 2 * N:
            wrapped = np.zeros(N, bool)
        assert np.array_equal(got[~wrapped], true[~wrapped]), k
        assert np.array_equal(got[wrapped], true[wrapped] - np.uint32(1)), k  # MAX - a = -a - 1 (Q1)
    # one CMUX step (trgsw.rs:174-196) chained from reference products: in1 + ExtProd(BSK[i], X^k*in1 - in1),
    # the external product assembled from Spqlios_poly_mul_1024 calls as in the test above
    sk, ck = keys128
    P = ck.params
    acc = rng.integers(0, 2**32, (2, N), dtype=np.uint64).astype(np.uint32)
    for i, k in ((3, 1), (500, N + 17), (699, 2 * N - 1)):
        rot = np.stack([O.poly_mul_with_x_k(acc[0], k), O.poly_mul_with_x_k(acc[1], k)])
        dec = O.decomposition(rot - acc, P.l, P.bgbit, ck.decomposition_offset)
        ext = np.zeros((2, N), np.uint32)
        for r in range(2 * P.l):
            for c in range(2):
                ext[c] += O.ref_poly_mul(ck.bootstrapping_key_time[i][r][c], dec[r])
        ref_cmux = acc + ext
        got = O.cmux(acc, rot, ck.bootstrapping_key[i], P.l, P.bgbit, ck.decomposition_offset)
        assert signed_diff(got, ref_cmux) <= 2 * P.l  # SPQLIOS truncation, one LSB per product
        acc = got  # chain
