"""A team of host threads calling the C ABI concurrently (rs-tfhe_amd/csrc/callers.cpp): the stand-in for a Rayon
team calling one `Send + Sync` strategy (src/bootstrap/mod.rs:23-38).  Load generator for the tests and for bench.py;
C++ threads, because Python threads would measure the interpreter lock."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _capi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libtfhe_callers.so")
_lib = None

OP_GATE, OP_BOOTSTRAP_LUT, OP_MUX = 0, 1, 2


class _Api(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("batch_gate", "batch_bootstrap", "batch_mux", "last_error", "pool_batch_gate",
                                          "pool_batch_bootstrap", "pool_batch_mux", "pool_last_error")]


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise ImportError(f"{_LIB_PATH} is missing: build it with `make -C rs-tfhe_amd/csrc`")
        _lib = C.CDLL(_LIB_PATH)
        _lib.tfhe_callers_run.restype = C.c_int
        _lib.tfhe_callers_run.argtypes = [C.POINTER(_Api), C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 6 + [
            C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_void_p, C.c_char_p, C.c_size_t]
    return _lib


def _api() -> _Api:
    h = _capi.lib()
    a = _Api()
    for name, _ in _Api._fields_:
        setattr(a, name, C.cast(getattr(h, "tfhe_hip_" + name), C.c_void_p))
    return a


def run(target, op: int, a, b=None, c=None, gates=None, testvecs=None, threads: int = 8, calls: int = 1,
        per_call: int = 1):
    """`threads` host threads, each making `calls` blocking calls of `per_call` ciphertexts on `target` (an Engine or
    a Pool); call i of thread t takes rows [(t * calls + i) * per_call, ...) of the operands.  Returns (out, seconds
    from the common start to the last return, per-call milliseconds [threads, calls])."""
    from .engine import Pool

    lib = _load()
    is_pool = isinstance(target, Pool)
    handle = target._h if is_pool else target._ctx
    total = threads * calls * per_call
    a = np.ascontiguousarray(a, dtype=np.uint32)
    assert a.ndim == 2 and a.shape[0] == total, (a.shape, total)
    width = a.shape[1]

    def prep(x, dtype=np.uint32):
        return None if x is None else np.ascontiguousarray(x, dtype=dtype)

    b, c, testvecs = prep(b), prep(c), prep(testvecs)
    gates = prep(gates, np.uint8)
    if gates is not None:
        assert gates.shape == (threads * calls,)
    if testvecs is not None:
        assert testvecs.shape == (threads * calls, 2, 1024)
    out = np.zeros_like(a)
    secs = C.c_double(0.0)
    call_ms = np.zeros((threads, calls), np.float64)
    err = C.create_string_buffer(512)
    api = _api()

    def p(x):
        return None if x is None else x.ctypes.data_as(C.c_void_p)

    rc = lib.tfhe_callers_run(C.byref(api), handle, int(is_pool), int(op), p(gates), p(a), p(b), p(c), p(testvecs), p(out),
                              width, per_call, threads, calls, C.byref(secs), p(call_ms), err, len(err))
    if rc != _capi.OK:
        raise _capi.TfheHipError(rc, err.value.decode())
    return out, secs.value, call_ms
