"""ctypes binding of the C ABI declared in include/tfhe_hip.h.

The product path has no CPU fallback: if libtfhe_hip.so is missing or cannot
be loaded this module raises, it never routes anywhere else.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TFHE_HIP_LIB", os.path.join(_HERE, "libtfhe_hip.so"))  # env: kernel A/B experiments only

OK, EINVAL, EHIP, ENOKEY, ENOMEM = 0, -1, -2, -3, -4


class TfheHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"tfhe_hip error {code}: {msg}")
        self.code = code


class Params(C.Structure):
    """struct tfhe_hip_params"""

    _fields_ = [("n", C.c_int32), ("l", C.c_int32), ("bgbit", C.c_int32), ("basebit", C.c_int32), ("t", C.c_int32)]


class KernelTimes(C.Structure):
    """struct tfhe_hip_kernel_times"""

    _fields_ = [
        ("blind_rotate_ms", C.c_double),
        ("key_switch_ms", C.c_double),
        ("blind_rotate_launches", C.c_uint64),
        ("key_switch_launches", C.c_uint64),
        ("bootstraps", C.c_uint64),
    ]


class ClockSample(C.Structure):
    """struct tfhe_hip_clock_sample"""

    _fields_ = [("shader_mhz", C.c_double), ("rtc_mhz", C.c_double), ("shader_cycles", C.c_uint64),
                ("rtc_ticks", C.c_uint64)]


class PoolTransferTimes(C.Structure):
    """struct tfhe_hip_pool_transfer_times"""

    _fields_ = [("scatter_ms_sum", C.c_double), ("scatter_ms_max", C.c_double), ("gather_ms_sum", C.c_double),
                ("gather_ms_max", C.c_double), ("scatter_bytes", C.c_uint64), ("gather_bytes", C.c_uint64),
                ("calls", C.c_uint64), ("comm_create_ms", C.c_double), ("key_replication_ms", C.c_double),
                ("scatter_group_ms_sum", C.c_double), ("gather_group_ms_sum", C.c_double)]


class CombineStats(C.Structure):
    """struct tfhe_hip_combine_stats"""

    _fields_ = [("max_count", C.c_uint64), ("launches", C.c_uint64), ("requests", C.c_uint64),
                ("ciphertexts", C.c_uint64), ("max_requests_per_launch", C.c_uint64), ("lingers", C.c_uint64),
                ("linger_us", C.c_double), ("pack_us", C.c_double), ("gpu_us", C.c_double), ("unpack_us", C.c_double)]


_P = C.c_void_p
_SZ = C.c_size_t
_CTX = C.c_void_p
_U32 = C.c_uint32

# name -> (restype, argtypes); every symbol include/tfhe_hip.h declares
SIGNATURES = {
    "tfhe_hip_ctx_create": (C.c_int, [C.POINTER(Params), C.c_int, C.POINTER(_CTX)]),
    "tfhe_hip_ctx_destroy": (None, [_CTX]),
    "tfhe_hip_last_error": (C.c_char_p, [_CTX]),
    "tfhe_hip_name": (C.c_char_p, []),
    "tfhe_hip_device_count": (C.c_int, []),
    "tfhe_hip_key_create": (C.c_int, [_CTX, C.POINTER(_CTX)]),
    "tfhe_hip_key_parent": (_CTX, [_CTX]),
    "tfhe_hip_key_is_loaded": (C.c_int, [_CTX]),
    "tfhe_hip_pool_key_create": (C.c_int, [_CTX, C.POINTER(_CTX)]),
    "tfhe_hip_load_cloud_key": (C.c_int, [_CTX, _P, _P, C.c_uint32, _P]),
    "tfhe_hip_gen_cloud_key": (C.c_int, [_CTX, _P, _P, C.c_double, C.c_double, C.c_uint64]),
    "tfhe_hip_gen_cloud_key_secure": (C.c_int, [_CTX, _P, _P, C.c_double, C.c_double]),
    "tfhe_hip_gen_cloud_key_with_key": (C.c_int, [_CTX, _P, _P, C.c_double, C.c_double, _P]),
    "tfhe_hip_cloud_key_buffers": (C.c_int, [_CTX, C.POINTER(_P), C.POINTER(_SZ), C.POINTER(_P), C.POINTER(_SZ),
                                             C.POINTER(_P), C.POINTER(_SZ), C.POINTER(C.c_uint32)]),
    "tfhe_hip_adopt_cloud_key": (C.c_int, [_CTX, C.c_uint32]),
    "tfhe_hip_host_alloc": (C.c_int, [_SZ, C.POINTER(_P)]),
    "tfhe_hip_host_free": (None, [_P]),
    "tfhe_hip_export_cloud_key": (C.c_int, [_CTX, _P, _P, C.POINTER(C.c_uint32), _P]),
    "tfhe_hip_batch_gate": (C.c_int, [_CTX, C.c_int, _P, _P, _P, _SZ]),
    "tfhe_hip_batch_gate_dev": (C.c_int, [_CTX, C.c_int, _P, _P, _P, _SZ, _P]),
    "tfhe_hip_batch_gates_mixed": (C.c_int, [_CTX, _P, _P, _P, _P, _SZ]),
    "tfhe_hip_batch_gates_mixed_dev": (C.c_int, [_CTX, _P, _P, _P, _P, _SZ, _P]),
    "tfhe_hip_batch_gates_mixed_nks": (C.c_int, [_CTX, _P, _P, _P, _P, _SZ]),
    "tfhe_hip_batch_gates_mixed_nks_dev": (C.c_int, [_CTX, _P, _P, _P, _P, _SZ, _P]),
    "tfhe_hip_batch_bootstrap": (C.c_int, [_CTX, _P, _P, C.c_int, C.c_int, _P, _SZ]),
    "tfhe_hip_batch_bootstrap_dev": (C.c_int, [_CTX, _P, _P, C.c_int, C.c_int, _P, _SZ, _P]),
    "tfhe_hip_batch_tlwe_lincomb": (C.c_int, [_CTX, C.c_uint32, _P, C.c_uint32, _P, C.c_uint32, _P, _SZ]),
    "tfhe_hip_batch_tlwe_lincomb_dev": (C.c_int, [_CTX, C.c_uint32, _P, C.c_uint32, _P, C.c_uint32, _P, _SZ, _P]),
    "tfhe_hip_batch_lincomb_bootstrap": (
        C.c_int, [_CTX, C.c_uint32, _P, C.c_uint32, _P, C.c_uint32, _P, C.c_int, C.c_int, _P, _SZ]),
    "tfhe_hip_batch_lincomb_bootstrap_dev": (
        C.c_int, [_CTX, C.c_uint32, _P, C.c_uint32, _P, C.c_uint32, _P, C.c_int, C.c_int, _P, _SZ, _P]),
    "tfhe_hip_batch_blind_rotate": (C.c_int, [_CTX, _P, _P, _P, _SZ]),
    "tfhe_hip_batch_blind_rotate_dev": (C.c_int, [_CTX, _P, _P, _P, _SZ, _P]),
    "tfhe_hip_batch_mux": (C.c_int, [_CTX, C.c_int, _P, _P, _P, _P, _SZ]),
    "tfhe_hip_batch_mux_dev": (C.c_int, [_CTX, C.c_int, _P, _P, _P, _P, _SZ, _P]),
    "tfhe_hip_batch_external_product": (C.c_int, [_CTX, _P, _P, _P, _SZ]),
    "tfhe_hip_batch_sample_extract": (C.c_int, [_CTX, _P, C.c_int, _P, _SZ]),
    "tfhe_hip_batch_identity_key_switch": (C.c_int, [_CTX, _P, _P, _SZ]),
    "tfhe_hip_load_reenc_key": (C.c_int, [_CTX, _P]),
    "tfhe_hip_reenc_key_is_loaded": (C.c_int, [_CTX]),
    "tfhe_hip_batch_reencrypt": (C.c_int, [_CTX, _P, _P, _SZ]),
    "tfhe_hip_batch_reencrypt_dev": (C.c_int, [_CTX, _P, _P, _SZ, _P]),
    "tfhe_hip_batch_ifft": (C.c_int, [_CTX, _P, _P, _SZ]),
    "tfhe_hip_batch_fft": (C.c_int, [_CTX, _P, _P, _SZ]),
    "tfhe_hip_batch_poly_mul": (C.c_int, [_CTX, _P, _P, _P, _SZ]),
    "tfhe_hip_set_profiling": (C.c_int, [_CTX, C.c_int]),
    "tfhe_hip_get_kernel_times": (C.c_int, [_CTX, C.POINTER(KernelTimes)]),
    "tfhe_hip_get_clock_sample": (C.c_int, [_CTX, C.POINTER(ClockSample)]),
    "tfhe_hip_get_key_switch_clock_sample": (C.c_int, [_CTX, C.POINTER(ClockSample)]),
    "tfhe_hip_synchronize": (C.c_int, [_CTX]),
    "tfhe_hip_set_combining": (C.c_int, [_CTX, _SZ]),
    "tfhe_hip_get_combine_stats": (C.c_int, [_CTX, C.POINTER(CombineStats)]),
    "tfhe_hip_describe_dispatch": (C.c_int, [_CTX, _SZ, C.c_char_p, _SZ]),
    "tfhe_hip_rounding_mode": (C.c_char_p, [_CTX]),
    # several GPUs behind one handle
    "tfhe_hip_pool_create": (C.c_int, [C.POINTER(Params), C.POINTER(C.c_int), C.c_int, C.POINTER(_CTX)]),
    "tfhe_hip_pool_destroy": (None, [_CTX]),
    "tfhe_hip_pool_size": (C.c_int, [_CTX]),
    "tfhe_hip_pool_members_for": (C.c_int, [_CTX, _SZ]),
    "tfhe_hip_pool_key_transport": (C.c_char_p, [_CTX]),
    "tfhe_hip_pool_ctx": (_CTX, [_CTX, C.c_int]),
    "tfhe_hip_pool_last_error": (C.c_char_p, [_CTX]),
    "tfhe_hip_pool_shard": (None, [_SZ, C.c_int, C.c_int, C.POINTER(_SZ), C.POINTER(_SZ)]),
    "tfhe_hip_pool_load_cloud_key": (C.c_int, [_CTX, _P, _P, C.c_uint32, _P]),
    "tfhe_hip_pool_gen_cloud_key_secure": (C.c_int, [_CTX, _P, _P, C.c_double, C.c_double]),
    "tfhe_hip_pool_gen_cloud_key_with_key": (C.c_int, [_CTX, _P, _P, C.c_double, C.c_double, _P]),
    "tfhe_hip_pool_gen_cloud_key": (C.c_int, [_CTX, _P, _P, C.c_double, C.c_double, C.c_uint64]),
    "tfhe_hip_pool_export_cloud_key": (C.c_int, [_CTX, C.c_int, _P, _P, C.POINTER(C.c_uint32), _P]),
    "tfhe_hip_pool_batch_gate": (C.c_int, [_CTX, C.c_int, _P, _P, _P, _SZ]),
    "tfhe_hip_pool_batch_gates_mixed": (C.c_int, [_CTX, _P, _P, _P, _P, _SZ]),
    "tfhe_hip_pool_batch_bootstrap": (C.c_int, [_CTX, _P, _P, C.c_int, C.c_int, _P, _SZ]),
    "tfhe_hip_pool_batch_mux": (C.c_int, [_CTX, C.c_int, _P, _P, _P, _P, _SZ]),
    "tfhe_hip_pool_batch_blind_rotate": (C.c_int, [_CTX, _P, _P, _P, _SZ]),
    "tfhe_hip_pool_batch_gates_mixed_nks": (C.c_int, [_CTX, _P, _P, _P, _P, _SZ]),
    "tfhe_hip_pool_batch_tlwe_lincomb": (C.c_int, [_CTX, _U32, _P, _U32, _P, _U32, _P, _SZ]),
    "tfhe_hip_pool_batch_lincomb_bootstrap": (C.c_int, [_CTX, _U32, _P, _U32, _P, _U32, _P, C.c_int, C.c_int, _P, _SZ]),
    # ... for a batch resident on one member's GPU (home_member, device pointers, stream)
    "tfhe_hip_pool_batch_gate_dev": (C.c_int, [_CTX, C.c_int, C.c_int, _P, _P, _P, _SZ, _P]),
    "tfhe_hip_pool_batch_gates_mixed_dev": (C.c_int, [_CTX, C.c_int, _P, _P, _P, _P, _SZ, _P]),
    "tfhe_hip_pool_batch_gates_mixed_nks_dev": (C.c_int, [_CTX, C.c_int, _P, _P, _P, _P, _SZ, _P]),
    "tfhe_hip_pool_batch_bootstrap_dev": (C.c_int, [_CTX, C.c_int, _P, _P, C.c_int, C.c_int, _P, _SZ, _P]),
    "tfhe_hip_pool_batch_tlwe_lincomb_dev": (C.c_int, [_CTX, C.c_int, _U32, _P, _U32, _P, _U32, _P, _SZ, _P]),
    "tfhe_hip_pool_batch_lincomb_bootstrap_dev": (
        C.c_int, [_CTX, C.c_int, _U32, _P, _U32, _P, _U32, _P, C.c_int, C.c_int, _P, _SZ, _P]),
    "tfhe_hip_pool_batch_mux_dev": (C.c_int, [_CTX, C.c_int, C.c_int, _P, _P, _P, _P, _SZ, _P]),
    "tfhe_hip_pool_batch_blind_rotate_dev": (C.c_int, [_CTX, C.c_int, _P, _P, _P, _SZ, _P]),
    "tfhe_hip_pool_synchronize": (C.c_int, [_CTX]),
    "tfhe_hip_pool_data_transport": (C.c_char_p, [_CTX]),
    "tfhe_hip_pool_set_profiling": (C.c_int, [_CTX, C.c_int]),
    "tfhe_hip_pool_get_transfer_times": (C.c_int, [_CTX, C.POINTER(PoolTransferTimes)]),
}

_lib = None


def lib() -> C.CDLL:
    """Load libtfhe_hip.so (built in-tree by __graft_entry__.build / csrc/Makefile)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C rs-tfhe_amd/csrc` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback."
            )
        # In a PyTorch process the HIP runtime torch bundles must be the one (and only) runtime
        # in the address space: load torch first so libtfhe_hip.so binds to it by SONAME.  (Two
        # HIP/HSA runtimes in one process leave the second without a device.)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        # a library built with the timing-only ablation switches (csrc/experiment.hpp) computes wrong results by
        # construction: only the experiment harnesses under profiles/exp/ may load it, and they say so
        if b"EXPERIMENT" in handle.tfhe_hip_name() and os.environ.get("TFHE_HIP_ALLOW_EXPERIMENT") != "1":
            raise ImportError(f"{LIB_PATH} is an ablated experiment build (results wrong by construction); "
                              "set TFHE_HIP_ALLOW_EXPERIMENT=1 only for timing runs under profiles/exp/")
        _lib = handle
    return _lib


def check(ctx, rc: int) -> None:
    if rc != OK:
        msg = lib().tfhe_hip_last_error(ctx)
        raise TfheHipError(rc, msg.decode() if msg else "")
