"""ctypes/numpy front-end of the CPU oracle (oracle/tfhe_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product path (rs-tfhe_amd/, libtfhe_hip.so)
never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libtfhe_oracle.so")
_REF_PATH = os.path.join(_HERE, "_ref", "libspqlios_ref.so")

N = 1024


def build(force: bool = False) -> None:
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "tfhe_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "libtfhe_oracle.so"])
    if os.path.isdir("/root/reference/src/fft/spqlios") and (force or not os.path.exists(_REF_PATH)):
        subprocess.check_call(["make", "-C", _HERE, "ref"])


class _Params(C.Structure):
    _fields_ = [
        ("n", C.c_int32),
        ("l", C.c_int32),
        ("bgbit", C.c_int32),
        ("basebit", C.c_int32),
        ("t", C.c_int32),
        ("alpha_lv0", C.c_double),
        ("alpha_lv1", C.c_double),
    ]


class _CloudKey(C.Structure):
    _fields_ = [
        ("P", _Params),
        ("decomposition_offset", C.c_uint32),
        ("testvec", C.c_void_p),
        ("bsk_fft", C.c_void_p),
        ("bsk_time", C.c_void_p),
        ("ksk", C.c_void_p),
    ]


@dataclass(frozen=True)
class Params:
    """Run-time form of the reference's SecurityParams (src/params.rs:53-84)."""

    name: str
    n: int
    l: int
    bgbit: int
    basebit: int
    t: int
    alpha_lv0: float
    alpha_lv1: float

    @property
    def base(self) -> int:
        return 1 << self.basebit

    def c(self) -> _Params:
        return _Params(self.n, self.l, self.bgbit, self.basebit, self.t, self.alpha_lv0, self.alpha_lv1)


# src/params.rs:91-116, 119-144, 148-173, 235-260, 379-404
SECURITY_80_BIT = Params("SECURITY_80_BIT", 550, 3, 6, 2, 7, 5.0e-5, 3.73e-8)
SECURITY_110_BIT = Params("SECURITY_110_BIT", 630, 3, 6, 2, 8, 3.0517578125e-05, 2.9802322387695313e-8)
SECURITY_128_BIT = Params("SECURITY_128_BIT", 700, 3, 6, 2, 9, 2.0e-5, 2.0e-8)
SECURITY_UINT1 = Params("SECURITY_UINT1", 700, 2, 10, 2, 8, 2.0e-05, 2.0e-08)
SECURITY_UINT4 = Params("SECURITY_UINT4", 820, 1, 22, 5, 3, 0.0000025167616095979554, 2.220446049250313e-16)
# src/params.rs:177-202, 206-231, 264-289, 322-347 (wider key-switch bases; used by the key-switch kernel tests)
SECURITY_UINT2 = Params("SECURITY_UINT2", 687, 1, 18, 4, 3, 0.00002120846893069972, 0.0000000000023184122752704995)
SECURITY_UINT3 = Params("SECURITY_UINT3", 820, 1, 23, 6, 2, 0.0000025167616095979554, 2.220446049250313e-16)
SECURITY_UINT5 = Params("SECURITY_UINT5", 1071, 1, 22, 6, 3, 7.08822676541043e-8, 2.2204460492503131e-17)
SECURITY_UINT6 = Params("SECURITY_UINT6", 1071, 1, 22, 6, 3, 7.08822676541043e-8, 2.2204460492503131e-17)  # params.rs:293-318
SECURITY_UINT7 = Params("SECURITY_UINT7", 1160, 1, 22, 7, 3, 1.9662200074984027e-8, 2.2204460492503131e-17)
SECURITY_UINT8 = Params("SECURITY_UINT8", 1160, 1, 22, 7, 3, 1.9662200074984027e-8, 2.2204460492503131e-17)  # params.rs:351-376
PARAM_SETS = {p.name: p for p in (SECURITY_80_BIT, SECURITY_110_BIT, SECURITY_128_BIT, SECURITY_UINT1, SECURITY_UINT2,
                                  SECURITY_UINT3, SECURITY_UINT4, SECURITY_UINT5, SECURITY_UINT6,
                                  SECURITY_UINT7, SECURITY_UINT8)}

# gate op codes -- shared with include/tfhe_hip.h
GATE_NAND, GATE_OR, GATE_AND, GATE_XOR, GATE_XNOR, GATE_NOR, GATE_ANDNY, GATE_ANDYN, GATE_ORNY, GATE_ORYN, GATE_COPY = range(11)
GATE_NAMES = ["nand", "or", "and", "xor", "xnor", "nor", "and_ny", "and_yn", "or_ny", "or_yn", "copy"]
# plaintext truth functions of the reference gates (gates.rs:54-150), as the reference's own
# unit tests assert them (gates.rs:559-653).  NB quirk Q8: Gates::xnor = a - 2b - 1/4 decrypts
# to XOR, and the reference's test_hom_xnor asserts exactly that (`false ^ (b ^ a)`,
# gates.rs:576-582) -- reproduced, not "fixed".
GATE_TRUTH = {
    GATE_NAND: lambda a, b: not (a and b),
    GATE_OR: lambda a, b: a or b,
    GATE_AND: lambda a, b: a and b,
    GATE_XOR: lambda a, b: a != b,
    GATE_XNOR: lambda a, b: False ^ (b ^ a),
    GATE_NOR: lambda a, b: not (a or b),
    GATE_ANDNY: lambda a, b: (not a) and b,
    GATE_ANDYN: lambda a, b: a and (not b),
    GATE_ORNY: lambda a, b: (not a) or b,
    GATE_ORYN: lambda a, b: a or (not b),
}

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_f64_to_torus.restype = C.c_uint32
        _lib.orc_f64_to_torus.argtypes = [C.c_double]
        _lib.orc_torus_to_f64.restype = C.c_double
        _lib.orc_torus_to_f64.argtypes = [C.c_uint32]
        _lib.orc_gen_decomposition_offset.restype = C.c_uint32
        _lib.orc_tlwe_phase.restype = C.c_uint32
        _lib.orc_lut_encode.restype = C.c_uint32
        _lib.orc_lwe_message_encoding.restype = C.c_double
        _lib.orc_lwe_message_encoding.argtypes = [C.c_int, C.c_int]
        _lib.orc_div_round.restype = C.c_size_t
        _lib.orc_div_round.argtypes = [C.c_size_t, C.c_size_t]
        _lib.orc_init()
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _u32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint32)


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


# ----------------------------------------------------------------------------
# scalar helpers
# ----------------------------------------------------------------------------
def f64_to_torus(d: float) -> int:
    return int(lib().orc_f64_to_torus(d))


def torus_to_f64(t: int) -> float:
    return float(lib().orc_torus_to_f64(C.c_uint32(t)))


def gen_decomposition_offset(l: int, bgbit: int) -> int:
    return int(lib().orc_gen_decomposition_offset(l, bgbit))


def gen_testvec() -> np.ndarray:
    tv = np.empty((2, N), np.uint32)
    lib().orc_gen_testvec(_p(tv))
    return tv


# ----------------------------------------------------------------------------
# FFT layer (klemsa.rs)
# ----------------------------------------------------------------------------
def cfft512(x: np.ndarray, inverse: bool = False) -> np.ndarray:
    re = _f64(x.real).copy()
    im = _f64(x.imag).copy()
    lib().orc_cfft512(_p(re), _p(im), int(inverse))
    return re + 1j * im


def klemsa_ifft(poly) -> np.ndarray:
    poly = _u32(poly)
    out = np.empty(N, np.float64)
    lib().orc_klemsa_ifft(_p(poly), _p(out))
    return out


def klemsa_fft(spec) -> np.ndarray:
    spec = _f64(spec)
    out = np.empty(N, np.uint32)
    lib().orc_klemsa_fft(_p(spec), _p(out))
    return out


def klemsa_poly_mul(a, b) -> np.ndarray:
    a, b = _u32(a), _u32(b)
    out = np.empty(N, np.uint32)
    lib().orc_klemsa_poly_mul(_p(a), _p(b), _p(out))
    return out


def negacyclic_schoolbook(a, b) -> np.ndarray:
    a, b = _u32(a), _u32(b)
    out = np.empty(N, np.uint32)
    lib().orc_negacyclic_schoolbook(_p(a), _p(b), _p(out))
    return out


# ----------------------------------------------------------------------------
# ring / LWE stages
# ----------------------------------------------------------------------------
def decomposition(trlwe, l: int, bgbit: int, offset: int) -> np.ndarray:
    trlwe = _u32(trlwe).reshape(2, N)
    out = np.empty((2 * l, N), np.uint32)
    lib().orc_decomposition(_p(trlwe), l, bgbit, C.c_uint32(offset), _p(out))
    return out


def poly_mul_with_x_k(a, k: int) -> np.ndarray:
    a = _u32(a)
    out = np.empty(N, np.uint32)
    lib().orc_poly_mul_with_x_k(_p(a), int(k), _p(out))
    return out


def external_product_fft(trgsw_fft, trlwe, l, bgbit, offset) -> np.ndarray:
    trgsw_fft = _f64(trgsw_fft)
    trlwe = _u32(trlwe)
    out = np.empty((2, N), np.uint32)
    lib().orc_external_product_fft(_p(trgsw_fft), _p(trlwe), l, bgbit, C.c_uint32(offset), _p(out))
    return out


def external_product_exact(trgsw_time, trlwe, l, bgbit, offset) -> np.ndarray:
    trgsw_time = _u32(trgsw_time)
    trlwe = _u32(trlwe)
    out = np.empty((2, N), np.uint32)
    lib().orc_external_product_exact(_p(trgsw_time), _p(trlwe), l, bgbit, C.c_uint32(offset), _p(out))
    return out


def cmux(in1, in2, cond_fft, l, bgbit, offset) -> np.ndarray:
    in1, in2, cond_fft = _u32(in1), _u32(in2), _f64(cond_fft)
    out = np.empty((2, N), np.uint32)
    lib().orc_cmux(_p(in1), _p(in2), _p(cond_fft), l, bgbit, C.c_uint32(offset), _p(out))
    return out


def sample_extract_index(trlwe, k: int = 0) -> np.ndarray:
    trlwe = _u32(trlwe)
    out = np.empty(N + 1, np.uint32)
    lib().orc_sample_extract_index(_p(trlwe), int(k), _p(out))
    return out


def sample_extract_index_2(trlwe, k: int, n: int) -> np.ndarray:
    trlwe = _u32(trlwe)
    out = np.empty(n + 1, np.uint32)
    lib().orc_sample_extract_index_2(_p(trlwe), int(k), int(n), _p(out))
    return out


def gate_prep(op: int, a, b, n: int) -> np.ndarray:
    a = _u32(a)
    out = np.empty(n + 1, np.uint32)
    bp = _p(_u32(b)) if b is not None else None
    rc = lib().orc_gate_prep(op, _p(a), bp, n, _p(out))
    if rc:
        raise ValueError(f"unknown gate op {op}")
    return out


# ----------------------------------------------------------------------------
# LUT (lut/encoder.rs, lut/generator.rs)
# ----------------------------------------------------------------------------
def lut_encode(message: int, m: int) -> int:
    return int(lib().orc_lut_encode(int(message), int(m)))


def lut_decode(value: int, m: int) -> int:
    return int(lib().orc_lut_decode(C.c_uint32(value), int(m)))


def lut_generate(f, m: int) -> np.ndarray:
    fvals = np.array([int(f(x)) for x in range(m)], dtype=np.int32)
    tv = np.empty((2, N), np.uint32)
    lib().orc_lut_generate(_p(fvals), int(m), _p(tv))
    return tv


def div_round(a: int, b: int) -> int:
    return int(lib().orc_div_round(a, b))


def lwe_message_encoding(message: int, m: int) -> float:
    return float(lib().orc_lwe_message_encoding(int(message), int(m)))


# ----------------------------------------------------------------------------
# keys and ciphertexts
# ----------------------------------------------------------------------------
class SecretKey:
    """src/key.rs:21-49"""

    def __init__(self, params: Params, seed: int):
        self.params = params
        self.key_lv0 = np.empty(params.n, np.uint32)
        self.key_lv1 = np.empty(N, np.uint32)
        lib().orc_gen_secret_key(C.c_uint64(seed), params.n, _p(self.key_lv0), _p(self.key_lv1))

    # tlwe.rs:37-58 / :84-98
    def encrypt_f64(self, p, seed: int) -> np.ndarray:
        p = _f64(np.atleast_1d(p))
        out = np.empty((len(p), self.params.n + 1), np.uint32)
        lib().orc_tlwe_encrypt_f64_batch(
            C.c_uint64(seed), _p(p), len(p), C.c_double(self.params.alpha_lv0), _p(self.key_lv0), self.params.n, _p(out)
        )
        return out

    def encrypt_bool(self, bits, seed: int) -> np.ndarray:
        bits = np.atleast_1d(np.asarray(bits)).astype(bool)
        return self.encrypt_f64(np.where(bits, 0.125, -0.125), seed)

    def encrypt_lwe_message(self, msgs, m: int, seed: int) -> np.ndarray:
        msgs = np.atleast_1d(np.asarray(msgs)).astype(np.int64) % m
        return self.encrypt_f64(msgs.astype(np.float64) * (1.0 / (2.0 * m)), seed)

    # tlwe.rs:60-68
    def decrypt_bool(self, cts) -> np.ndarray:
        cts = _u32(cts).reshape(-1, self.params.n + 1)
        return np.array(
            [bool(lib().orc_tlwe_decrypt_bool(_p(ct), _p(self.key_lv0), self.params.n)) for ct in cts]
        )

    def phase(self, cts) -> np.ndarray:
        cts = _u32(cts).reshape(-1, self.params.n + 1)
        s = self.key_lv0.astype(np.uint32)
        inner = (cts[:, :-1] * s[None, :]).sum(axis=1, dtype=np.uint32)
        return (cts[:, -1] - inner).astype(np.uint32)

    # tlwe.rs:111-126
    def decrypt_lwe_message(self, cts, m: int) -> np.ndarray:
        cts = _u32(cts).reshape(-1, self.params.n + 1)
        return np.array(
            [int(lib().orc_tlwe_decrypt_lwe_message(_p(ct), int(m), _p(self.key_lv0), self.params.n)) for ct in cts]
        )

    def decrypt_bool_lv1(self, ct_lv1) -> bool:
        ct_lv1 = _u32(ct_lv1)
        return bool(lib().orc_tlwe_decrypt_bool(_p(ct_lv1), _p(self.key_lv1), N))

    def trlwe_phase(self, trlwe) -> np.ndarray:
        """b - a (*) s1 (exact, trlwe.rs:69-81 without the sign test)."""
        trlwe = _u32(trlwe).reshape(2, N)
        return (trlwe[1] - negacyclic_schoolbook(trlwe[0], self.key_lv1)).astype(np.uint32)


class CloudKey:
    """src/key.rs:51-66 -- generated by the oracle's own seeded keygen."""

    def __init__(self, sk: SecretKey, seed: int, with_time_domain: bool = False):
        P = sk.params
        self.params = P
        self.decomposition_offset = gen_decomposition_offset(P.l, P.bgbit)
        self.blind_rotate_testvec = gen_testvec()
        self.bootstrapping_key = np.empty((P.n, 2 * P.l, 2, N), np.float64)
        self.bootstrapping_key_time = np.empty((P.n, 2 * P.l, 2, N), np.uint32) if with_time_domain else None
        self.key_switching_key = np.empty((N, P.t, P.base, P.n + 1), np.uint32)
        cp = P.c()
        lib().orc_gen_bootstrapping_key(
            C.c_uint64(seed * 2 + 1),
            C.byref(cp),
            _p(sk.key_lv0),
            _p(sk.key_lv1),
            _p(self.bootstrapping_key),
            _p(self.bootstrapping_key_time) if with_time_domain else None,
        )
        lib().orc_gen_key_switching_key(
            C.c_uint64(seed * 2 + 2), C.byref(cp), _p(sk.key_lv0), _p(sk.key_lv1), _p(self.key_switching_key)
        )

    @classmethod
    def from_arrays(cls, params: Params, bootstrapping_key, key_switching_key, decomposition_offset,
                    blind_rotate_testvec) -> "CloudKey":
        """Wrap key material produced elsewhere (e.g. exported from the GPU key generator) so the CPU
        path can be run under the very same key."""
        self = cls.__new__(cls)
        self.params = params
        self.decomposition_offset = int(decomposition_offset)
        self.blind_rotate_testvec = np.ascontiguousarray(blind_rotate_testvec, np.uint32).reshape(2, N)
        self.bootstrapping_key = np.ascontiguousarray(bootstrapping_key, np.float64).reshape(
            params.n, 2 * params.l, 2, N)
        self.bootstrapping_key_time = None
        self.key_switching_key = np.ascontiguousarray(key_switching_key, np.uint32).reshape(
            N, params.t, params.base, params.n + 1)
        return self

    def c(self) -> _CloudKey:
        ck = _CloudKey()
        ck.P = self.params.c()
        ck.decomposition_offset = self.decomposition_offset
        ck.testvec = self.blind_rotate_testvec.ctypes.data
        ck.bsk_fft = self.bootstrapping_key.ctypes.data
        ck.bsk_time = self.bootstrapping_key_time.ctypes.data if self.bootstrapping_key_time is not None else None
        ck.ksk = self.key_switching_key.ctypes.data
        return ck


def keygen(params: Params, seed: int, with_time_domain: bool = False):
    sk = SecretKey(params, seed)
    return sk, CloudKey(sk, seed, with_time_domain)


# ----------------------------------------------------------------------------
# whole-path functions
# ----------------------------------------------------------------------------
def blind_rotate(ck: CloudKey, ct, testvec=None, exact: bool = False) -> np.ndarray:
    ct = _u32(ct)
    tv = _u32(testvec if testvec is not None else ck.blind_rotate_testvec)
    out = np.empty((2, N), np.uint32)
    cp = ck.params.c()
    if exact:
        assert ck.bootstrapping_key_time is not None
        lib().orc_blind_rotate_exact(
            _p(ct), _p(tv), _p(ck.bootstrapping_key_time), C.byref(cp), C.c_uint32(ck.decomposition_offset), _p(out)
        )
    else:
        lib().orc_blind_rotate(
            _p(ct), _p(tv), _p(ck.bootstrapping_key), C.byref(cp), C.c_uint32(ck.decomposition_offset), _p(out)
        )
    return out


def identity_key_switching(ck: CloudKey, ct_lv1) -> np.ndarray:
    ct_lv1 = _u32(ct_lv1)
    out = np.empty(ck.params.n + 1, np.uint32)
    cp = ck.params.c()
    lib().orc_identity_key_switching(_p(ct_lv1), _p(ck.key_switching_key), C.byref(cp), _p(out))
    return out


def batch_identity_key_switching(ck: CloudKey, cts_lv1) -> np.ndarray:
    """identity_key_switching (trgsw.rs:332-360) mapped over [count][N+1]; the C calls run on a few host threads
    (ctypes releases the GIL) -- the tests' counts reach four figures."""
    from concurrent.futures import ThreadPoolExecutor

    cts_lv1 = _u32(cts_lv1).reshape(-1, N + 1)
    out = np.empty((len(cts_lv1), ck.params.n + 1), np.uint32)
    cp = ck.params.c()
    ksk = ck.key_switching_key
    fn = lib().orc_identity_key_switching

    def one(i):
        fn(_p(cts_lv1[i]), _p(ksk), C.byref(cp), _p(out[i]))

    with ThreadPoolExecutor(max(1, min(16, os.cpu_count() or 1))) as ex:
        list(ex.map(one, range(len(cts_lv1))))
    return out


# ----------------------------------------------------------------------------
# proxy re-encryption: src/proxy_reenc.rs (feature `proxy-reenc`)
# ----------------------------------------------------------------------------
class PublicKeyLv0:
    """proxy_reenc.rs:95-222: `size` encryptions of zero ([size][n+1]) and public-key encryption with them."""

    def __init__(self, sk: SecretKey, seed: int, size: int = 0, alpha: float = None):
        self.params = sk.params
        size = 2 * sk.params.n if not size else int(size)  # :125-131
        alpha = sk.params.alpha_lv0 if alpha is None else alpha
        self.encryptions = np.empty((size, sk.params.n + 1), np.uint32)
        lib().orc_gen_public_key_lv0(C.c_uint64(seed), _p(sk.key_lv0), sk.params.n, size, C.c_double(alpha), _p(self.encryptions))

    def encrypt_f64(self, p, alpha: float, seed: int) -> np.ndarray:
        p = _f64(np.atleast_1d(p))
        out = np.empty((len(p), self.params.n + 1), np.uint32)
        lib().orc_public_key_encrypt_f64_batch(C.c_uint64(seed), _p(self.encryptions), len(self.encryptions), self.params.n,
                                                _p(p), len(p), C.c_double(alpha), _p(out))
        return out

    def encrypt_bool(self, bits, alpha: float, seed: int) -> np.ndarray:
        bits = np.atleast_1d(np.asarray(bits)).astype(bool)
        return self.encrypt_f64(np.where(bits, 0.125, -0.125), alpha, seed)


def gen_reenc_key(params: Params, key_from, seed: int, key_to=None, public_key_to: PublicKeyLv0 = None,
                  alpha: float = None) -> np.ndarray:
    """ProxyReencryptionKey::new_symmetric[_with_params] (proxy_reenc.rs:362-425; pass key_to = the target level-0 key)
    or ::new_asymmetric[_with_params] (:271-330; pass public_key_to).  Returns key_encryptions [n*t*base][n+1]; base
    and t are the parameter set's."""
    assert (key_to is None) != (public_key_to is None)
    alpha = params.alpha_lv0 if alpha is None else alpha  # params::KSK_ALPHA (params.rs:468)
    key = np.empty((params.n * params.t * params.base, params.n + 1), np.uint32)
    cp = params.c()
    kf = _u32(key_from)
    if public_key_to is None:
        kt = _u32(key_to)
        lib().orc_gen_reenc_key(C.c_uint64(seed), C.byref(cp), _p(kf), _p(kt), None, 0, C.c_double(alpha), _p(key))
    else:
        pk = public_key_to.encryptions
        lib().orc_gen_reenc_key(C.c_uint64(seed), C.byref(cp), _p(kf), None, _p(pk), len(pk), C.c_double(alpha), _p(key))
    return key


def reencrypt_tlwe_lv0(params: Params, key_encryptions, cts) -> np.ndarray:
    """reencrypt_tlwe_lv0 (proxy_reenc.rs:468-510) over [count][n+1]."""
    cts = _u32(cts).reshape(-1, params.n + 1)
    key = _u32(key_encryptions)
    out = np.empty_like(cts)
    cp = params.c()
    lib().orc_batch_reencrypt(_p(cts), _p(key), C.byref(cp), _p(out), C.c_size_t(len(cts)))
    return out


def batch_gate(ck: CloudKey, op: int, a, b, nthreads: int = 0) -> np.ndarray:
    a = _u32(a).reshape(-1, ck.params.n + 1)
    bb = _u32(b).reshape(-1, ck.params.n + 1) if b is not None else None
    out = np.empty_like(a)
    cck = ck.c()
    rc = lib().orc_batch_gate(C.byref(cck), int(op), _p(a), _p(bb) if bb is not None else None, _p(out), len(a), nthreads)
    if rc:
        raise ValueError(f"unknown gate op {op}")
    return out


def batch_bootstrap(ck: CloudKey, cts, testvec=None, keyswitch: bool = True, nthreads: int = 0) -> np.ndarray:
    cts = _u32(cts).reshape(-1, ck.params.n + 1)
    out = np.empty_like(cts)
    per_ct = 0
    tvp = None
    if testvec is not None:
        testvec = _u32(testvec)
        per_ct = int(testvec.ndim == 3)
        tvp = _p(testvec)
    cck = ck.c()
    lib().orc_batch_bootstrap(C.byref(cck), _p(cts), tvp, per_ct, int(keyswitch), _p(out), len(cts), nthreads)
    return out


def batch_blind_rotate(ck: CloudKey, cts, testvec=None, nthreads: int = 0) -> np.ndarray:
    cts = _u32(cts).reshape(-1, ck.params.n + 1)
    out = np.empty((len(cts), 2, N), np.uint32)
    tvp = _p(_u32(testvec)) if testvec is not None else None
    cck = ck.c()
    lib().orc_batch_blind_rotate(C.byref(cck), _p(cts), tvp, _p(out), len(cts), nthreads)
    return out


def batch_mux(ck: CloudKey, a, b, c, naive: bool, nthreads: int = 0) -> np.ndarray:
    a = _u32(a).reshape(-1, ck.params.n + 1)
    b = _u32(b).reshape(-1, ck.params.n + 1)
    c = _u32(c).reshape(-1, ck.params.n + 1)
    out = np.empty_like(a)
    cck = ck.c()
    lib().orc_batch_mux(C.byref(cck), int(naive), _p(a), _p(b), _p(c), _p(out), len(a), nthreads)
    return out


def num_threads() -> int:
    return int(lib().orc_num_threads())


# ----------------------------------------------------------------------------
# reference SPQLIOS build (oracle/_ref) -- present only where /root/reference was
# ----------------------------------------------------------------------------
_ref = None


def ref_available() -> bool:
    return os.path.exists(_REF_PATH)


def ref_lib():
    """The reference's own C ABI: src/fft/spqlios/spqlios-wrapper.cpp:10-41."""
    global _ref
    if _ref is None:
        _ref = C.CDLL(_REF_PATH)
        _ref.Spqlios_new.restype = C.c_void_p
        _ref.Spqlios_new.argtypes = [C.c_int32]
        _ref.Spqlios_poly_mul_1024.argtypes = [C.c_void_p] * 4
        _ref.Spqlios_ifft_lv1.argtypes = [C.c_void_p] * 3
        _ref.Spqlios_fft_lv1.argtypes = [C.c_void_p] * 3
        _ref._handle = _ref.Spqlios_new(N)
    return _ref


def ref_poly_mul(a, b) -> np.ndarray:
    r = ref_lib()
    a, b = _u32(a), _u32(b)
    out = np.empty(N, np.uint32)
    r.Spqlios_poly_mul_1024(r._handle, _p(out), _p(a), _p(b))
    return out


def ref_roundtrip(a) -> np.ndarray:
    r = ref_lib()
    a = _u32(a)
    spec = np.empty(N, np.float64)
    out = np.empty(N, np.uint32)
    r.Spqlios_ifft_lv1(r._handle, _p(spec), _p(a))
    r.Spqlios_fft_lv1(r._handle, _p(out), _p(spec))
    return out
