"""A second, independent restatement of the composed path in plain numpy.

TEST INFRASTRUCTURE ONLY (like the rest of oracle/).  It exists to pin the C
restatement (tfhe_oracle.c): it was written separately, straight from the
reference's formulas, shares no code with the C file, and uses numpy's own FFT
for the 512-point complex DFT the reference takes from rustfft.  tests/
test_oracle_numpy_model.py requires the two to agree word for word on whole
bootstraps at bgbit = 6, where the f64 product is exact after rounding, so
neither FFT implementation's round-off can show.

Every function names the reference lines (under /root/reference/) it follows.
N = 1024 throughout; (n, l, bgbit, basebit, t) are run-time values.
"""
from __future__ import annotations

import numpy as np

N = 1024
N2 = 512
_TWIST = np.exp(1j * np.pi * np.arange(N2) / N)  # src/fft/klemsa.rs:49-58: exp(i*pi*k/N)


def u32(x) -> np.ndarray:
    return np.asarray(x).astype(np.uint32)


def klemsa_ifft(poly) -> np.ndarray:
    """src/fft/klemsa.rs:88-117: signed fold, twist, forward DFT, x2; returns re || im (1024 f64)."""
    p = u32(poly).view(np.int32).astype(np.float64)
    z = (p[:N2] + 1j * p[N2:]) * _TWIST
    f = np.fft.fft(z) * 2.0
    return np.concatenate([f.real, f.imag])


def klemsa_fft(spec) -> np.ndarray:
    """src/fft/klemsa.rs:119-150: x0.5, unnormalised inverse DFT, conj twist, /512, round, wrap."""
    s = np.asarray(spec, np.float64)
    f = np.fft.ifft((s[:N2] + 1j * s[N2:]) * 0.5) * N2  # numpy's ifft divides by n; rustfft's does not
    t = f * np.conj(_TWIST) * (1.0 / N2)
    # f64::round (half away from zero), then `as i64 as u32`
    re = np.trunc(t.real + np.copysign(0.5, t.real)).astype(np.int64)
    im = np.trunc(t.imag + np.copysign(0.5, t.imag)).astype(np.int64)
    return np.concatenate([re, im]).astype(np.uint32)


def fma_in_fd(res, a, b) -> None:
    """src/trgsw.rs:118-142: res += 0.5 * a * b per complex bin, in the reference's association."""
    res[:N2] = (a[N2:] * b[N2:]) * 0.5 - res[:N2]
    res[:N2] = (a[:N2] * b[:N2]) * 0.5 - res[:N2]
    res[N2:] += (a[:N2] * b[N2:] + a[N2:] * b[:N2]) * 0.5


def decomposition(a, b, l: int, bgbit: int, offset: int) -> np.ndarray:
    """src/trgsw.rs:144-171: rows 0..l from a, l..2l from b; digits stored as wrapped u32."""
    mask = np.uint32((1 << bgbit) - 1)
    half = np.uint32(1 << (bgbit - 1))
    out = np.empty((2 * l, N), np.uint32)
    for src, base in ((u32(a) + np.uint32(offset), 0), (u32(b) + np.uint32(offset), l)):
        for i in range(l):
            out[base + i] = ((src >> np.uint32(32 - (i + 1) * bgbit)) & mask) - half
    return out


def poly_mul_with_x_k(a, k: int) -> np.ndarray:
    """src/trgsw.rs:307-330, k in [0, 2N]; wrapped coefficients are MAX - a[i] (not -a[i])."""
    a = u32(a)
    res = np.zeros(N, np.uint32)
    mx = np.uint32(0xFFFFFFFF)
    if k < N:
        res[k:] = a[: N - k]
        res[:k] = mx - a[N - k:]
    else:
        res[k - N:] = mx - a[: 2 * N - k]
        res[: k - N] = a[2 * N - k:]
    return res


def external_product(bsk_i, a, b, l: int, bgbit: int, offset: int):
    """src/trgsw.rs:77-116.  bsk_i: [2l][2][N] f64 (one TRGSWLv1FFT)."""
    dec = decomposition(a, b, l, bgbit, offset)
    out_a = np.zeros(N)
    out_b = np.zeros(N)
    for r in range(2 * l):
        d = klemsa_ifft(dec[r])
        fma_in_fd(out_a, d, bsk_i[r, 0])
        fma_in_fd(out_b, d, bsk_i[r, 1])
    return klemsa_fft(out_a), klemsa_fft(out_b)


def blind_rotate(ct, bsk, testvec, l: int, bgbit: int, offset: int):
    """src/trgsw.rs:198-226 (and :242-274 with a caller's test vector); cmux :174-196 inlined."""
    ct = u32(ct)
    n = len(ct) - 1
    b_tilda = 2 * N - ((int(ct[n]) + (1 << 20)) >> 21)  # usize add: no 32-bit wrap
    ra = poly_mul_with_x_k(testvec[0], b_tilda)
    rb = poly_mul_with_x_k(testvec[1], b_tilda)
    for i in range(n):
        a_tilda = ((int(ct[i]) + (1 << 20)) & 0xFFFFFFFF) >> 21  # wrapping_add in u32
        xa, xb = poly_mul_with_x_k(ra, a_tilda), poly_mul_with_x_k(rb, a_tilda)
        ea, eb = external_product(bsk[i], xa - ra, xb - rb, l, bgbit, offset)
        ra, rb = ea + ra, eb + rb
    return ra, rb


def sample_extract_index0(ra, rb) -> np.ndarray:
    """src/trlwe.rs:106-120 with k = 0."""
    out = np.empty(N + 1, np.uint32)
    out[0] = ra[0]
    out[1:N] = np.uint32(0xFFFFFFFF) - ra[N - 1:0:-1]
    out[N] = rb[0]
    return out


def identity_key_switching(lv1, ksk, n: int, basebit: int, t: int) -> np.ndarray:
    """src/trgsw.rs:332-360.  ksk: [N][t][base][n+1] u32."""
    lv1 = u32(lv1)
    res = np.zeros(n + 1, np.uint32)
    res[n] = lv1[N]
    prec = np.uint32(1 << (32 - (1 + basebit * t)))
    mask = (1 << basebit) - 1
    for i in range(N):
        a_bar = int(lv1[i] + prec)
        for j in range(t):
            k = (a_bar >> (32 - (j + 1) * basebit)) & mask
            if k:
                res -= ksk[i, j, k]
    return res


# src/gates.rs:54-150: (ca, cb, const) with prepared = ca*a + cb*b, prepared.b += f64_to_torus(const)
GATE_PREP = {
    "nand": (-1, -1, 0.125), "or": (1, 1, 0.125), "and": (1, 1, -0.125), "xor": (1, 2, 0.25),
    "xnor": (1, -2, -0.25), "nor": (-1, -1, -0.125),
}


def f64_to_torus(d: float) -> int:
    """src/utils.rs:9-12."""
    import math

    return int(math.fmod(d, 1.0) * 4294967296.0) & 0xFFFFFFFF


def gate(name: str, ca_ct, cb_ct, bsk, ksk, testvec, n, l, bgbit, basebit, t, offset) -> np.ndarray:
    """One gate end to end: prep, blind rotate, sample extract, identity key switch."""
    ca, cb, cst = GATE_PREP[name]
    x = u32(ca_ct) * np.uint32(ca & 0xFFFFFFFF) + u32(cb_ct) * np.uint32(cb & 0xFFFFFFFF)
    x[n:] += np.array([f64_to_torus(cst)], np.uint32)  # array add: wraps silently, as wrapping_add does
    ra, rb = blind_rotate(x, bsk, testvec, l, bgbit, offset)
    return identity_key_switching(sample_extract_index0(ra, rb), ksk, n, basebit, t)
