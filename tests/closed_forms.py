"""Closed forms that tie two integer stages to executed reference code / to arithmetic no restatement can share a slip
with (used by tests/test_oracle_closed_forms.py on the CPU checker and tests/test_gpu_closed_forms.py on the HIP
kernels).

sample_extract_index (src/trlwe.rs:106-120).  For a TRLWE (a, b) under the ring key s, the sample extracted at index k
must carry coefficient k of the phase polynomial b - a (*) s.  The reference "negates" with Torus::MAX - x (quirk Q1:
-x - 1), so every wrapped position i > k contributes one extra s_i:
    phase(extract_k) = (b - a (*) s)[k] + #{ i > k : s_i = 1 }                                   (mod 2^32)
with a (*) s computed by the REFERENCE's own compiled Spqlios_poly_mul_1024 (oracle/_ref).  SPQLIOS truncates its
output (fft_processor_spqlios.cpp:128-129: +-1 LSB), so `a` holds multiples of 256: a (*) s is then a multiple of 256 and
rounding the reference product to the nearest one recovers it exactly.

identity_key_switching (src/trgsw.rs:332-360) under a NOISE-FREE key-switching key (alpha = 0 in gen_key_switching_key,
src/key.rs:102-122: row (i, j, k) encrypts k * s1_i / 2^((j+1) basebit) exactly):
    phase(out) = src.b - sum_i s1_i * trunc_{t basebit}(a_i + PREC_OFFSET)                       (mod 2^32)
where trunc keeps the top t * basebit bits and PREC_OFFSET = 2^(32 - (1 + basebit t)).  This sees what a decrypt test
cannot: PREC_OFFSET, the digit order, the row index base*t*i + base*j + k and the sign.
"""
import ctypes as C

import numpy as np

N = 1024


def round256(x):
    return ((x.astype(np.uint64) + 128) // 256 * 256).astype(np.uint32)


def extract_phase_expected(O, trlwe, key_lv1):
    """[N] u32: for every k, what the phase of sample_extract_index(trlwe, k) must be (a = multiples of 256)."""
    a, b = trlwe
    assert not (a & 255).any()
    prod = round256(O.ref_poly_mul(a, key_lv1.astype(np.uint32)))
    phase_poly = (b - prod).astype(np.uint32)
    s = key_lv1.astype(np.int64)
    ones_above = (s.sum() - np.cumsum(s)).astype(np.uint32)  # [k] = #{i > k : s_i = 1}
    return (phase_poly + ones_above).astype(np.uint32)


def lv1_phase(lv1, key_lv1):
    """b - <a, s1> of [count][N+1] level-1 samples"""
    lv1 = np.asarray(lv1, np.uint32).reshape(-1, N + 1)
    inner = (lv1[:, :N] * key_lv1.astype(np.uint32)[None, :]).sum(axis=1, dtype=np.uint32)
    return (lv1[:, N] - inner).astype(np.uint32)


def noise_free_ksk(O, params, sk, seed=99):
    """orc_gen_key_switching_key (key.rs:102-122) with alpha = 0: [N][t][base][n+1]"""
    import dataclasses

    P0 = dataclasses.replace(params, alpha_lv0=0.0)
    ksk = np.empty((N, params.t, params.base, params.n + 1), np.uint32)
    cp = P0.c()
    O.lib().orc_gen_key_switching_key(C.c_uint64(seed), C.byref(cp), sk.key_lv0.ctypes.data_as(C.c_void_p),
                                      sk.key_lv1.ctypes.data_as(C.c_void_p), ksk.ctypes.data_as(C.c_void_p))
    return ksk


def key_switch_phase_expected(params, lv1, key_lv1, prec_offset=None):
    """[count] u32: the phase the key-switched samples must have under key_lv0, for a noise-free key-switching key"""
    lv1 = np.asarray(lv1, np.uint32).reshape(-1, N + 1)
    bits = params.basebit * params.t
    if prec_offset is None:
        prec_offset = 1 << (32 - (1 + bits))
    abar = (lv1[:, :N] + np.uint32(prec_offset)).astype(np.uint32)
    trunc = abar & np.uint32((0xFFFFFFFF << (32 - bits)) & 0xFFFFFFFF)
    inner = (trunc * key_lv1.astype(np.uint32)[None, :]).sum(axis=1, dtype=np.uint32)
    return (lv1[:, N] - inner).astype(np.uint32)


# ---- the composed bootstrap on TRIVIAL ciphertexts: a closed form that needs no key ---------------------------------------
# blind_rotate (src/trgsw.rs:198-226) starts from X^b~ * testvec with b~ = 2N - ((b + 2^20) >> 21) (a non-wrapping add,
# quirk Q2) and then walks the mask; for a == 0 every rotation amount is 0, every CMUX step decomposes the zero polynomial
# (all digits 0, since the offset puts Bg/2 in every digit position) and leaves the accumulator alone -- EXACTLY, whatever
# the key.  sample_extract_index_2(., 0) then carries coefficient 0 of X^b~ * testvec.b in its last word.  From
# poly_mul_with_x_k (src/trgsw.rs:307-330), with r = (b + 2^20) >> 21 in [0, 2N]:
#     r == 0 or r == 2N: p[0];    0 < r < N: p[r];    r == N: MAX - p[0];    N < r < 2N: MAX - p[r - N]   (quirk Q1)
# For a lookup table of f at message modulus m (src/lut/generator.rs:89-137) and b = x / (2m) + delta, |delta| inside the
# half-slot, that coefficient is encode(f(x)) = f64_to_torus((f(x) mod m) / (2m)) (src/lut/encoder.rs:66-73) -- except for
# x = 0 and delta < 0, where the rotation wraps: the table's negated tail comes back through MAX - v = -v - 1, ONE LSB below
# encode(f(0)).  This pins, through the real bootstrap entry points: b~ (Q2), the direction of the rotation, Q1, the
# table's layout (slot width, half-slot offset, negated tail) and the encoder -- with no secret key and no noise.
def trivial_ciphertexts(n, phases):
    cts = np.zeros((len(phases), n + 1), np.uint32)
    cts[:, n] = np.asarray(phases, np.uint64).astype(np.uint32)
    return cts


def lut_trivial_cases(f, m, margin=1 << 22):
    """(phases, expected last words) for every message x of [0, m) at several places of its slot"""
    phases, expect = [], []
    half = (1 << 32) // (4 * m)  # half a slot, in torus units
    for x in range(m):
        centre = (x << 32) // (2 * m)
        enc = ((int(f(x)) % m) << 32) // (2 * m)  # f64_to_torus(y / (2m)) for y < m: exact
        for delta in (-(half - margin), -(half // 2), -1, 0, 1, half // 2, half - margin):
            phase = (centre + delta) % (1 << 32)
            phases.append(phase)
            r = (phase + (1 << 20)) >> 21  # (phase just below 2^32 rounds to r = 2N: X^0, no wrap -- the non-wrapping add of Q2)
            wraps = 1024 <= r < 2048
            assert not wraps or (x == 0 and delta < 0)
            expect.append((enc - 1) % (1 << 32) if wraps else enc)
    return np.array(phases, np.uint64), np.array(expect, np.uint64).astype(np.uint32)


def gate_testvec_trivial_cases():
    """the cloud key's own test vector (b == 1/8 everywhere, src/key.rs:91-100): +1/8 where the rotation does not wrap,
    MAX - 1/8 where it does"""
    phases = np.array([0, 1, 1 << 20, (1 << 31) - (1 << 20) - 1, (1 << 31) - (1 << 20), 1 << 31, (3 << 30), (1 << 32) - (1 << 20) - 1,
                       (1 << 32) - (1 << 20), (1 << 32) - 1], np.uint64)
    r = (phases + (1 << 20)) >> 21
    expect = np.where((r >= 1024) & (r < 2048), 0xFFFFFFFF - 0x20000000, 0x20000000).astype(np.uint32)
    return phases, expect


# ---- gate prep through the composed bootstrap, no key ------------------------------------------------------------------------
# The linear forms of src/gates.rs:54-150 (read off the reference: coefficient of a, coefficient of b, constant in eighths of
# the torus).  On trivial inputs (zero masks, phases pa / pb) the prepared sample is trivial with phase ca*pa + cb*pb + cc, and
# the bootstrap without key switch under the cloud key's own test vector returns +1/8 where the rotation does not wrap and
# MAX - 1/8 where it does (gate_testvec_trivial_cases above): every sample is one exact constraint on (ca, cb, cc).
GATE_FORMS = {  # tfhe_hip_gate code -> (ca, cb, cc / 2^29)
    0: (-1, -1, +1),  # nand   gates.rs:54-58    -(a + b) + 1/8
    1: (+1, +1, +1),  # or     :62-66             a + b + 1/8
    2: (+1, +1, -1),  # and    :70-74             a + b - 1/8
    3: (+1, +2, +2),  # xor    :78-82             a + 2b + 1/4
    4: (+1, -2, -2),  # xnor   :86-90             a - 2b - 1/4
    5: (-1, -1, -1),  # nor    :94-98            -(a + b) - 1/8
    6: (-1, +1, -1),  # and_ny :102-111          -a + b - 1/8
    7: (+1, -1, -1),  # and_yn :115-124           a - b - 1/8
    8: (-1, +1, +1),  # or_ny  :128-137          -a + b + 1/8
    9: (+1, -1, +1),  # or_yn  :141-150           a - b + 1/8
}


def gate_trivial_expected(gate, pa, pb):
    ca, cb, cc = GATE_FORMS[int(gate)]
    phase = (ca * pa.astype(np.int64) + cb * pb.astype(np.int64) + cc * (1 << 29)) % (1 << 32)
    r = (phase + (1 << 20)) >> 21
    return np.where((r >= 1024) & (r < 2048), 0xFFFFFFFF - 0x20000000, 0x20000000).astype(np.uint32)


def trivial_mask_expected(n, phases):
    """The first n words of the bootstrap-without-key-switch of trivial ciphertexts, for ANY test vector with a == 0:
    X^b~ * 0 is MAX on the coefficients that wrap and 0 elsewhere (quirk Q1; poly_mul_with_x_k, trgsw.rs:307-330: k < N wraps
    [0, k), N <= k < 2N wraps [k - N, N), k == 2N nothing), and sample_extract_index_2(., 0) (trlwe.rs:122-136, with its
    N := n) reads p[0] = a[0], p[i] = MAX - a[n - i]."""
    phases = np.asarray(phases, np.uint64)
    k = 2 * N - ((phases + (1 << 20)) >> 21).astype(np.int64)  # b~ in [0, 2N]
    j = np.arange(N)[None, :]
    kk = k[:, None]
    wrapped = np.where(kk < N, j < kk, (j >= kk - N) & (kk < 2 * N))
    a_rot = np.where(wrapped, 0xFFFFFFFF, 0).astype(np.uint32)  # [count][N]
    out = np.empty((len(phases), n), np.uint32)
    out[:, 0] = a_rot[:, 0]
    idx = n - np.arange(1, n)
    out[:, 1:] = (np.uint32(0xFFFFFFFF) - a_rot[:, idx]).astype(np.uint32)
    return out
