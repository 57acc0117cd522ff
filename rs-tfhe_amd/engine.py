"""Engine: one C-ABI context = one GPU (one process per GPU under torch.distributed)."""
from __future__ import annotations

import ctypes as C
import threading
from typing import Optional

import numpy as np

from . import _capi
from .params import N, SecurityParams

# gate selectors (enum tfhe_hip_gate)
NAND, OR, AND, XOR, XNOR, NOR, ANDNY, ANDYN, ORNY, ORYN, COPY = range(11)
GATE_IDS = {
    "nand": NAND, "or": OR, "and": AND, "xor": XOR, "xnor": XNOR, "nor": NOR,
    "and_ny": ANDNY, "and_yn": ANDYN, "or_ny": ORNY, "or_yn": ORYN, "copy": COPY,
}


def _u32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint32)


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _is_tensor(x) -> bool:
    return type(x).__module__.startswith("torch")


def _tptr(t):
    """Device pointer of a contiguous int32/uint32 CUDA tensor."""
    if t is None:
        return None
    if not t.is_cuda or not t.is_contiguous() or t.element_size() != 4:
        raise ValueError("device tensors must be contiguous 32-bit CUDA tensors")
    return C.c_void_p(t.data_ptr())


def pinned_empty(shape, dtype=np.uint32) -> np.ndarray:
    """A numpy array in pinned host memory (`tfhe_hip_host_alloc`).  The host entry points read and write such
    arrays in place over PCIe -- no staging copies -- when every ciphertext operand of the call is pinned."""
    lib = _capi.lib()
    shape = (shape,) if np.isscalar(shape) else tuple(shape)
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = C.c_void_p()
    rc = lib.tfhe_hip_host_alloc(max(nbytes, 1), C.byref(p))
    if rc != _capi.OK:
        msg = lib.tfhe_hip_last_error(None)
        raise _capi.TfheHipError(rc, msg.decode() if msg else "")
    buf = (C.c_uint8 * max(nbytes, 1)).from_address(p.value)
    arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
    import weakref

    weakref.finalize(buf, lib.tfhe_hip_host_free, C.c_void_p(p.value))  # freed when the last view is gone
    return arr


def _out_like(a: np.ndarray, out) -> np.ndarray:
    if out is None:
        return np.empty_like(a)
    if out.dtype != np.uint32 or out.shape != a.shape or not out.flags.c_contiguous:
        raise ValueError("out must be a C-contiguous uint32 array of the operands' shape")
    return out


def pinned_copy(a) -> np.ndarray:
    a = np.ascontiguousarray(a)
    out = pinned_empty(a.shape, a.dtype)
    out[...] = a
    return out


def device_count() -> int:
    """GPUs this process can open (`tfhe_hip_device_count`): what `Pool(params, range(device_count()))` spans."""
    return int(_capi.lib().tfhe_hip_device_count())


class Engine:
    """Owns a tfhe_hip_ctx.  Host arrays are numpy uint32; *_dev methods take
    torch CUDA tensors (int32 storage of the u32 words) and only enqueue work."""

    def __init__(self, params: SecurityParams, device: int = 0, _view_of: "Engine" = None):
        self.params = params
        self.device = device
        self._lib = _capi.lib()
        ctx = C.c_void_p()
        if _view_of is not None:  # a key view: another resident cloud key on the parent's context
            rc = self._lib.tfhe_hip_key_create(_view_of._ctx, C.byref(ctx))
            if rc != _capi.OK:
                raise _capi.TfheHipError(rc, "tfhe_hip_key_create failed")
        else:
            cp = _capi.Params(params.n, params.l, params.bgbit, params.basebit, params.iks_t)
            rc = self._lib.tfhe_hip_ctx_create(C.byref(cp), device, C.byref(ctx))
            if rc != _capi.OK:
                msg = self._lib.tfhe_hip_last_error(None)
                raise _capi.TfheHipError(rc, msg.decode() if msg else "")
        self._ctx = ctx
        self._owner = None  # a Pool when the context is borrowed from one (tfhe_hip_pool_ctx): never destroyed here
        self._parent = _view_of  # keeps the parent context alive for as long as this view exists
        self._views = []  # weak references to the live key views of this context (closed before it)
        self._key = None  # the CloudKey object currently loaded (held, so identity cannot be recycled)
        self.lock = threading.RLock()  # for callers that want several calls on this handle back to back
        self._last_use = 0
        self._users = 0  # bootstrap.keyed_engine: calls in flight under this view (never evicted while > 0)
        if _view_of is not None:
            import weakref

            _view_of._views.append(weakref.ref(self))

    def new_key_view(self) -> "Engine":
        """Another resident cloud key on this context (`tfhe_hip_key_create`): an Engine handle with its own key
        that shares this context's device, streams, scratch buffers and mutex.  Every method works on it; calls
        under different views may come from different threads.  Replaces the reference's `&CloudKey` argument
        (src/bootstrap/mod.rs:23-38): a call names its key by the handle it is made on."""
        base = self._parent if self._parent is not None else self
        return Engine(base.params, base.device, _view_of=base)

    @classmethod
    def from_pool(cls, pool: "Pool", member: int) -> "Engine":
        """The member context of a pool as an Engine (for the device-resident `*_dev` entry points).  The pool keeps
        ownership; do not run pool batch calls while this engine has work in flight (include/tfhe_hip.h)."""
        ctx = pool._lib.tfhe_hip_pool_ctx(pool._h, int(member))
        if not ctx:
            raise ValueError("no such pool member")
        self = cls.__new__(cls)
        self.params, self.device, self._lib = pool.params, pool.devices[member], pool._lib
        self._ctx = C.c_void_p(ctx)
        self._owner = pool
        self._parent = None
        self._views = []
        self._key = ("pool", object())
        self.lock = threading.RLock()
        self._last_use = 0
        self._users = 0
        return self

    # -- lifetime -------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_ctx", None):
            for ref in getattr(self, "_views", []):  # key views go before the context they run on
                v = ref()
                if v is not None:
                    v.close()
            self._views = []
            if getattr(self, "_owner", None) is None:
                self._lib.tfhe_hip_ctx_destroy(self._ctx)
            self._ctx = None
            self._owner = None
            self._parent = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc: int) -> None:
        _capi.check(self._ctx, rc)

    @property
    def name(self) -> str:
        return self._lib.tfhe_hip_name().decode()

    # -- cloud key ------------------------------------------------------------
    def load_cloud_key(self, cloud_key) -> None:
        """cloud_key: any object with the reference CloudKey fields (src/key.rs:51-56):
        decomposition_offset, blind_rotate_testvec [2][N], key_switching_key
        [N][t][base][n+1], bootstrapping_key [n][2l][2][N] f64."""
        p = self.params
        bsk = np.ascontiguousarray(cloud_key.bootstrapping_key, dtype=np.float64)
        ksk = _u32(cloud_key.key_switching_key)
        tv = _u32(cloud_key.blind_rotate_testvec)
        if bsk.size != p.n * 2 * p.l * 2 * N:
            raise ValueError("bootstrapping_key has the wrong size for these parameters")
        if ksk.size != N * p.iks_t * p.base * (p.n + 1):
            raise ValueError("key_switching_key has the wrong size for these parameters")
        if tv.size != 2 * N:
            raise ValueError("blind_rotate_testvec must be [2][N]")
        self._chk(
            self._lib.tfhe_hip_load_cloud_key(
                self._ctx, _ptr(bsk), _ptr(ksk), C.c_uint32(int(cloud_key.decomposition_offset)), _ptr(tv)
            )
        )
        self._key = cloud_key

    def gen_cloud_key(self, key_lv0, key_lv1, seed=None, alpha_ksk=None, alpha_bsk=None, rng_key: bytes = None) -> None:
        """CloudKey::new(&secret_key) (src/key.rs:59-66) on the GPU, straight into this context.

        seed=None (default): masks and noise come from a ChaCha20 stream keyed by the operating system's CSPRNG
        (`tfhe_hip_gen_cloud_key_secure`), or by the caller's 32-byte `rng_key`.  An integer `seed` makes the key
        reproducible and as guessable as the seed: tests and benchmarks only (include/tfhe_hip.h)."""
        p = self.params
        k0, k1 = _u32(key_lv0).reshape(-1), _u32(key_lv1).reshape(-1)
        if len(k0) != p.n or len(k1) != N:
            raise ValueError("secret key has the wrong size for these parameters")
        a0 = C.c_double(p.alpha_lv0 if alpha_ksk is None else alpha_ksk)
        a1 = C.c_double(p.alpha_lv1 if alpha_bsk is None else alpha_bsk)
        if rng_key is not None:
            if seed is not None or len(rng_key) != 32:
                raise ValueError("rng_key is 32 bytes and excludes seed")
            buf = (C.c_uint8 * 32).from_buffer_copy(bytes(rng_key))
            self._chk(self._lib.tfhe_hip_gen_cloud_key_with_key(self._ctx, _ptr(k0), _ptr(k1), a0, a1, C.addressof(buf)))
        elif seed is None:
            self._chk(self._lib.tfhe_hip_gen_cloud_key_secure(self._ctx, _ptr(k0), _ptr(k1), a0, a1))
        else:
            self._chk(self._lib.tfhe_hip_gen_cloud_key(self._ctx, _ptr(k0), _ptr(k1), a0, a1,
                                                       C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF)))
        self._key = ("generated", object())

    def export_cloud_key(self):
        """The context's key back as a CloudKey in the reference layouts."""
        from .key import CloudKey

        p = self.params
        bsk = np.empty((p.n, 2 * p.l, 2, N), np.float64)
        ksk = np.empty((N, p.iks_t, p.base, p.n + 1), np.uint32)
        tv = np.empty((2, N), np.uint32)
        off = C.c_uint32(0)
        self._chk(self._lib.tfhe_hip_export_cloud_key(self._ctx, _ptr(bsk), _ptr(ksk), C.byref(off), _ptr(tv)))
        return CloudKey(p, bsk, ksk, int(off.value), tv)

    def cloud_key_device_tensors(self):
        """(bsk, ksk, testvec, decomposition_offset): the context's key buffers in the engine layouts as uint8 torch
        tensors that ALIAS them (no copy), for device-to-device replication (`distributed.broadcast_engine_key`)."""
        import torch

        ptrs = [C.c_void_p() for _ in range(3)]
        sizes = [C.c_size_t() for _ in range(3)]
        off = C.c_uint32(0)
        self._chk(self._lib.tfhe_hip_cloud_key_buffers(self._ctx, C.byref(ptrs[0]), C.byref(sizes[0]), C.byref(ptrs[1]),
                                                       C.byref(sizes[1]), C.byref(ptrs[2]), C.byref(sizes[2]), C.byref(off)))

        class _Alias:  # CUDA array interface: torch wraps the pointer without copying
            def __init__(self, ptr, nbytes):
                self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}

        dev = torch.device("cuda", self.device)
        ts = [torch.as_tensor(_Alias(int(p.value), int(n.value)), device=dev) for p, n in zip(ptrs, sizes)]
        return ts[0], ts[1], ts[2], int(off.value)

    def adopt_cloud_key(self, decomposition_offset: int) -> None:
        """The key buffers were filled from outside (see cloud_key_device_tensors): make them the current key."""
        self._chk(self._lib.tfhe_hip_adopt_cloud_key(self._ctx, C.c_uint32(int(decomposition_offset))))
        self._key = ("adopted", object())

    def ensure_key(self, cloud_key) -> None:
        if self._key is not cloud_key:
            self.load_cloud_key(cloud_key)

    # -- batched hot path, host arrays -----------------------------------------
    def _cts(self, a) -> np.ndarray:
        a = _u32(a)
        return a.reshape(-1, self.params.n + 1)

    def batch_gate(self, gate: int, a, b=None, out=None) -> np.ndarray:
        """`out`: optional preallocated [count][n+1] uint32 result array -- pass pinned arrays (`pinned_empty`) for
        a, b and out and the call runs without staging copies."""
        a = self._cts(a)
        bb = self._cts(b) if b is not None else None
        if bb is not None and bb.shape != a.shape:
            raise ValueError("operand batches differ in shape")
        out = _out_like(a, out)
        self._chk(self._lib.tfhe_hip_batch_gate(self._ctx, int(gate), _ptr(a), _ptr(bb), _ptr(out), len(a)))
        return out

    def batch_gates_mixed(self, gates, a, b, keyswitch: bool = True) -> np.ndarray:
        """Per-ciphertext gate selectors (one launch for a whole circuit level)."""
        a, b = self._cts(a), self._cts(b)
        g = np.ascontiguousarray(gates, dtype=np.uint8).reshape(-1)
        if len(g) != len(a) or b.shape != a.shape:
            raise ValueError("gates / operand batches differ in length")
        out = np.empty_like(a)
        fn = self._lib.tfhe_hip_batch_gates_mixed if keyswitch else self._lib.tfhe_hip_batch_gates_mixed_nks
        self._chk(fn(self._ctx, _ptr(g), _ptr(a), _ptr(b), _ptr(out), len(a)))
        return out

    def batch_bootstrap(self, cts, testvec=None, keyswitch: bool = True) -> np.ndarray:
        cts = self._cts(cts)
        out = np.empty_like(cts)
        per_ct = 0
        tv = None
        if testvec is not None:
            tv = _u32(testvec)
            per_ct = int(tv.ndim == 3)
            if tv.size != (len(cts) if per_ct else 1) * 2 * N:
                raise ValueError("test vector must be [2][N], or [count][2][N] for per-ciphertext tables")
        self._chk(
            self._lib.tfhe_hip_batch_bootstrap(self._ctx, _ptr(cts), _ptr(tv), per_ct, int(keyswitch), _ptr(out), len(cts))
        )
        return out

    def batch_tlwe_lincomb(self, ca: int, a, cb: int = 0, b=None, cconst: int = 0) -> np.ndarray:
        """ca*a + cb*b on every word, + cconst on the body: TLWE Add / Sub / Neg / AddMul / SubMul
        (src/tlwe.rs:129-214)."""
        a = self._cts(a)
        bb = self._cts(b) if b is not None else None
        if (cb & 0xFFFFFFFF) and (bb is None or bb.shape != a.shape):
            raise ValueError("second operand missing or of a different shape")
        out = np.empty_like(a)
        self._chk(self._lib.tfhe_hip_batch_tlwe_lincomb(
            self._ctx, ca & 0xFFFFFFFF, _ptr(a), cb & 0xFFFFFFFF, _ptr(bb), cconst & 0xFFFFFFFF, _ptr(out), len(a)))
        return out

    def batch_lincomb_bootstrap(self, ca: int, a, cb: int = 0, b=None, cconst: int = 0, testvec=None,
                                keyswitch: bool = True) -> np.ndarray:
        """bootstrap(ca*a + cb*b + cconst) with an optional LookupTable.poly: the combination is formed in
        the prologue of the blind-rotation kernel (examples/lut_add_two_numbers.rs:124-158)."""
        a = self._cts(a)
        bb = self._cts(b) if b is not None else None
        if (cb & 0xFFFFFFFF) and (bb is None or bb.shape != a.shape):
            raise ValueError("second operand missing or of a different shape")
        tv, per_ct = None, 0
        if testvec is not None:
            tv = _u32(testvec)
            per_ct = int(tv.ndim == 3)
            if tv.size != (len(a) if per_ct else 1) * 2 * N:
                raise ValueError("test vector must be [2][N], or [count][2][N] for per-ciphertext tables")
        out = np.empty_like(a)
        self._chk(self._lib.tfhe_hip_batch_lincomb_bootstrap(
            self._ctx, ca & 0xFFFFFFFF, _ptr(a), cb & 0xFFFFFFFF, _ptr(bb), cconst & 0xFFFFFFFF, _ptr(tv), per_ct,
            int(keyswitch), _ptr(out), len(a)))
        return out

    def batch_blind_rotate(self, cts, testvec=None) -> np.ndarray:
        cts = self._cts(cts)
        out = np.empty((len(cts), 2, N), np.uint32)
        tv = _u32(testvec) if testvec is not None else None
        if tv is not None and tv.size != 2 * N:
            raise ValueError("test vector must be [2][N]")
        self._chk(self._lib.tfhe_hip_batch_blind_rotate(self._ctx, _ptr(cts), _ptr(tv), _ptr(out), len(cts)))
        return out

    def batch_mux(self, a, b, c, naive: bool) -> np.ndarray:
        a, b, c = self._cts(a), self._cts(b), self._cts(c)
        if b.shape != a.shape or c.shape != a.shape:
            raise ValueError("operand batches differ in shape")
        out = np.empty_like(a)
        self._chk(self._lib.tfhe_hip_batch_mux(self._ctx, int(naive), _ptr(a), _ptr(b), _ptr(c), _ptr(out), len(a)))
        return out

    # -- single stages ----------------------------------------------------------
    def batch_external_product(self, trlwe, bsk_index) -> np.ndarray:
        trlwe = _u32(trlwe).reshape(-1, 2, N)
        idx = np.ascontiguousarray(bsk_index, dtype=np.int32).reshape(-1)
        if len(idx) != len(trlwe):
            raise ValueError("one bootstrapping-key index per TRLWE sample")
        out = np.empty_like(trlwe)
        self._chk(self._lib.tfhe_hip_batch_external_product(self._ctx, _ptr(trlwe), _ptr(idx), _ptr(out), len(trlwe)))
        return out

    def batch_sample_extract(self, trlwe, k: int = 0) -> np.ndarray:
        trlwe = _u32(trlwe).reshape(-1, 2, N)
        out = np.empty((len(trlwe), N + 1), np.uint32)
        self._chk(self._lib.tfhe_hip_batch_sample_extract(self._ctx, _ptr(trlwe), int(k), _ptr(out), len(trlwe)))
        return out

    def batch_identity_key_switch(self, lv1) -> np.ndarray:
        lv1 = _u32(lv1).reshape(-1, N + 1)
        out = np.empty((len(lv1), self.params.n + 1), np.uint32)
        self._chk(self._lib.tfhe_hip_batch_identity_key_switch(self._ctx, _ptr(lv1), _ptr(out), len(lv1)))
        return out

    # -- proxy re-encryption (src/proxy_reenc.rs; rs-tfhe_amd/proxy_reenc.py holds the client side) --------
    def load_reenc_key(self, key_encryptions) -> None:
        """ProxyReencryptionKey::key_encryptions [n][t][base][n+1] (proxy_reenc.rs:224-233) -> this handle (a context
        or a key view holds EITHER a cloud key OR a re-encryption key: `tfhe_hip_load_reenc_key`)."""
        p = self.params
        key = _u32(key_encryptions)
        if key.size != p.n * p.iks_t * p.base * (p.n + 1):
            raise ValueError("re-encryption key has the wrong size for these parameters")
        self._chk(self._lib.tfhe_hip_load_reenc_key(self._ctx, _ptr(key)))

    def reenc_key_is_loaded(self) -> bool:
        return self._lib.tfhe_hip_reenc_key_is_loaded(self._ctx) == 1  # 0 / 1; anything else is not "loaded"

    def batch_reencrypt(self, cts) -> np.ndarray:
        """proxy_reenc::reencrypt_tlwe_lv0 (proxy_reenc.rs:468-510) over [count][n+1] host ciphertexts."""
        cts = _u32(cts).reshape(-1, self.params.n + 1)
        out = np.empty_like(cts)
        self._chk(self._lib.tfhe_hip_batch_reencrypt(self._ctx, _ptr(cts), _ptr(out), len(cts)))
        return out

    def batch_reencrypt_dev(self, a, out, stream=None) -> None:
        """The same on int32 CUDA tensors [count][n+1] of this engine's GPU; only enqueues."""
        count = self._dev_batch(a, out)
        self._chk(self._lib.tfhe_hip_batch_reencrypt_dev(self._ctx, self._tp(a), self._tp(out), count, self._stream_ptr(stream)))

    def batch_ifft(self, polys) -> np.ndarray:
        polys = _u32(polys).reshape(-1, N)
        out = np.empty((len(polys), N), np.float64)
        self._chk(self._lib.tfhe_hip_batch_ifft(self._ctx, _ptr(out), _ptr(polys), len(polys)))
        return out

    def batch_fft(self, spectra) -> np.ndarray:
        spectra = np.ascontiguousarray(spectra, dtype=np.float64).reshape(-1, N)
        out = np.empty((len(spectra), N), np.uint32)
        self._chk(self._lib.tfhe_hip_batch_fft(self._ctx, _ptr(out), _ptr(spectra), len(spectra)))
        return out

    def batch_poly_mul(self, a, b) -> np.ndarray:
        a, b = _u32(a).reshape(-1, N), _u32(b).reshape(-1, N)
        if b.shape != a.shape:
            raise ValueError("operand batches differ in shape")
        out = np.empty_like(a)
        self._chk(self._lib.tfhe_hip_batch_poly_mul(self._ctx, _ptr(out), _ptr(a), _ptr(b), len(a)))
        return out

    # -- device-resident path (torch CUDA tensors; enqueue only) ------------------
    def _stream_ptr(self, stream):
        if stream is None:
            import torch

            stream = torch.cuda.current_stream(self.device)  # of THIS engine's GPU, whatever torch's current device is
        elif getattr(stream, "device", None) is not None and stream.device.index != self.device:
            raise ValueError(f"stream lives on {stream.device}, the engine on cuda:{self.device}")
        # torch's default stream is the legacy null stream (handle 0); the C ABI reads NULL as "the
        # context's own stream", so name the null stream explicitly: hipStreamLegacy == (hipStream_t)1
        return C.c_void_p(stream.cuda_stream or 1)

    def _tp(self, t):
        """Device pointer of a tensor that must live on this engine's GPU (test vectors, outputs, gate codes)."""
        p = _tptr(t)
        if t is not None and t.device.index != self.device:
            raise ValueError(f"device tensor lives on {t.device}, the engine on cuda:{self.device}")
        return p

    def _dev_batch(self, *tensors, width=None) -> int:
        """All tensors are [count][width] on this engine's GPU; returns count."""
        width = self.params.n + 1 if width is None else width
        first = tensors[0]
        for t in tensors:
            if t is None:
                continue
            if t.dim() != 2 or t.shape[1] != width or t.shape[0] != first.shape[0]:
                raise ValueError(f"device tensors must all be [count][{width}]")
            if t.device.index != self.device:
                raise ValueError(f"device tensor lives on {t.device}, the engine on cuda:{self.device}")
        return first.shape[0]

    def batch_gate_dev(self, gate: int, a, b, out, stream=None) -> None:
        count = self._dev_batch(a, b, out)
        self._chk(
            self._lib.tfhe_hip_batch_gate_dev(self._ctx, int(gate), self._tp(a), self._tp(b), self._tp(out), count, self._stream_ptr(stream))
        )

    def batch_gates_mixed_dev(self, gates, a, b, out, stream=None, keyswitch: bool = True) -> None:
        """gates: uint8 CUDA tensor [count]; a, b, out: int32 CUDA tensors [count][n+1].  keyswitch=False ends
        in bootstrap_without_key_switch (the first level of Gates::mux, `tfhe_hip_batch_gates_mixed_nks_dev`)."""
        if not gates.is_cuda or gates.element_size() != 1 or not gates.is_contiguous():
            raise ValueError("gates must be a contiguous uint8 CUDA tensor")
        if gates.device.index != self.device:
            raise ValueError("gates tensor lives on another device than this engine")
        count = self._dev_batch(a, b, out)
        if gates.numel() != count:
            raise ValueError("one gate code per ciphertext")
        fn = self._lib.tfhe_hip_batch_gates_mixed_dev if keyswitch else self._lib.tfhe_hip_batch_gates_mixed_nks_dev
        self._chk(fn(self._ctx, C.c_void_p(gates.data_ptr()), self._tp(a), self._tp(b), self._tp(out), count, self._stream_ptr(stream)))

    def batch_bootstrap_dev(self, cts, out, testvec=None, per_ct: bool = False, keyswitch: bool = True, stream=None) -> None:
        count = self._dev_batch(cts, out)
        if testvec is not None and testvec.numel() != (count if per_ct else 1) * 2 * N:
            raise ValueError("test vector must be [2][N], or [count][2][N] with per_ct")
        self._chk(
            self._lib.tfhe_hip_batch_bootstrap_dev(
                self._ctx, self._tp(cts), self._tp(testvec), int(per_ct), int(keyswitch), self._tp(out), count, self._stream_ptr(stream)
            )
        )

    def batch_tlwe_lincomb_dev(self, ca: int, a, cb: int, b, cconst: int, out, stream=None) -> None:
        count = self._dev_batch(a, b, out)
        self._chk(self._lib.tfhe_hip_batch_tlwe_lincomb_dev(
            self._ctx, ca & 0xFFFFFFFF, self._tp(a), cb & 0xFFFFFFFF, self._tp(b), cconst & 0xFFFFFFFF, self._tp(out), count,
            self._stream_ptr(stream)))

    def batch_lincomb_bootstrap_dev(self, ca: int, a, cb: int, b, cconst: int, out, testvec=None,
                                    per_ct: bool = False, keyswitch: bool = True, stream=None) -> None:
        count = self._dev_batch(a, b, out)
        if testvec is not None and testvec.numel() != (count if per_ct else 1) * 2 * N:
            raise ValueError("test vector must be [2][N], or [count][2][N] with per_ct")
        self._chk(self._lib.tfhe_hip_batch_lincomb_bootstrap_dev(
            self._ctx, ca & 0xFFFFFFFF, self._tp(a), cb & 0xFFFFFFFF, self._tp(b), cconst & 0xFFFFFFFF, self._tp(testvec),
            int(per_ct), int(keyswitch), self._tp(out), count, self._stream_ptr(stream)))

    def batch_blind_rotate_dev(self, cts, out_trlwe, testvec=None, stream=None) -> None:
        count = self._dev_batch(cts)
        if out_trlwe.numel() != count * 2 * N or (testvec is not None and testvec.numel() != 2 * N):
            raise ValueError("out_trlwe must be [count][2][N], testvec [2][N]")
        self._chk(
            self._lib.tfhe_hip_batch_blind_rotate_dev(
                self._ctx, self._tp(cts), self._tp(testvec), self._tp(out_trlwe), count, self._stream_ptr(stream)
            )
        )

    def batch_mux_dev(self, a, b, c, out, naive: bool, stream=None) -> None:
        count = self._dev_batch(a, b, c, out)
        self._chk(
            self._lib.tfhe_hip_batch_mux_dev(
                self._ctx, int(naive), self._tp(a), self._tp(b), self._tp(c), self._tp(out), count, self._stream_ptr(stream)
            )
        )

    # -- measurement ------------------------------------------------------------
    def set_profiling(self, enabled: bool) -> None:
        self._chk(self._lib.tfhe_hip_set_profiling(self._ctx, int(enabled)))

    def kernel_times(self) -> dict:
        kt = _capi.KernelTimes()
        self._chk(self._lib.tfhe_hip_get_kernel_times(self._ctx, C.byref(kt)))
        return {
            "blind_rotate_ms": kt.blind_rotate_ms,
            "key_switch_ms": kt.key_switch_ms,
            "blind_rotate_launches": int(kt.blind_rotate_launches),
            "key_switch_launches": int(kt.key_switch_launches),
            "bootstraps": int(kt.bootstraps),
        }

    def clock_sample(self) -> dict:
        """Shader clock the blind-rotation kernel actually ran at (sampled while profiling is on)."""
        cs = _capi.ClockSample()
        self._chk(self._lib.tfhe_hip_get_clock_sample(self._ctx, C.byref(cs)))
        return {"shader_mhz": cs.shader_mhz, "rtc_mhz": cs.rtc_mhz, "shader_cycles": int(cs.shader_cycles),
                "rtc_ticks": int(cs.rtc_ticks)}

    def key_switch_clock_sample(self) -> dict:
        """The same sample for the matrix-core key switch (k_key_switch_mfma)."""
        cs = _capi.ClockSample()
        self._chk(self._lib.tfhe_hip_get_key_switch_clock_sample(self._ctx, C.byref(cs)))
        return {"shader_mhz": cs.shader_mhz, "rtc_mhz": cs.rtc_mhz, "shader_cycles": int(cs.shader_cycles),
                "rtc_ticks": int(cs.rtc_ticks)}

    def synchronize(self) -> None:
        self._chk(self._lib.tfhe_hip_synchronize(self._ctx))

    # -- concurrent callers -------------------------------------------------------
    def set_combining(self, max_count: int) -> None:
        """Host-pointer calls of at most `max_count` ciphertexts made by concurrent threads share launches
        (`tfhe_hip_set_combining`; the default is the device's CU count, 0 switches it off)."""
        self._chk(self._lib.tfhe_hip_set_combining(self._ctx, int(max_count)))

    def combine_stats(self) -> dict:
        """Counters of the combining front end since the last call (`tfhe_hip_get_combine_stats`)."""
        st = _capi.CombineStats()
        self._chk(self._lib.tfhe_hip_get_combine_stats(self._ctx, C.byref(st)))
        return {k: (float(getattr(st, k)) if k.endswith("_us") else int(getattr(st, k))) for k, _ in st._fields_}

    @property
    def rounding_mode(self) -> str:
        """"fast" / "general": the blind-rotation kernels' rounding of the external product (`tfhe_hip_rounding_mode`)."""
        return self._lib.tfhe_hip_rounding_mode(self._ctx).decode()

    def describe_dispatch(self, count: int) -> str:
        """Which kernels a batch of `count` runs on (`tfhe_hip_describe_dispatch`), e.g.
        "blind_rotate=batch[0,1024)+single[1024,1100) key_switch=mfma(k=4)".  Needs the key loaded."""
        buf = C.create_string_buffer(256)
        self._chk(self._lib.tfhe_hip_describe_dispatch(self._ctx, int(count), buf, len(buf)))
        return buf.value.decode()


class Pool:
    """Several GPUs behind one handle (`tfhe_hip_pool`): the reference's Rayon `par_map` over the ciphertexts of
    a batch (src/parallel/rayon_impl.rs:40-47) as a map over devices.  `devices` may repeat an index (two
    contexts on one GPU).  The cloud key goes to the first device once and is replicated device to device;
    every batch call splits its host arrays contiguously over the members and keeps input order."""

    def __init__(self, params: SecurityParams, devices, _view_of: "Pool" = None):
        self.params = params
        self.devices = [int(d) for d in devices]
        self._lib = _capi.lib()
        h = C.c_void_p()
        if _view_of is not None:  # a key view of a pool: one key view per member context
            rc = self._lib.tfhe_hip_pool_key_create(_view_of._h, C.byref(h))
            if rc != _capi.OK:
                raise _capi.TfheHipError(rc, "tfhe_hip_pool_key_create failed")
        else:
            cp = _capi.Params(params.n, params.l, params.bgbit, params.basebit, params.iks_t)
            arr = (C.c_int * len(self.devices))(*self.devices)
            rc = self._lib.tfhe_hip_pool_create(C.byref(cp), arr, len(self.devices), C.byref(h))
            if rc != _capi.OK:
                msg = self._lib.tfhe_hip_pool_last_error(None)
                raise _capi.TfheHipError(rc, msg.decode() if msg else "")
        self._h = h
        self.home = 0  # member whose GPU holds the operands of the *_dev calls (their `home` argument's default)
        self._parent = _view_of  # keeps the parent pool alive for as long as this view exists
        self._views = []
        if _view_of is not None:
            import weakref

            _view_of._views.append(weakref.ref(self))

    def new_key_view(self) -> "Pool":
        """Another resident cloud key on every member of this pool (`tfhe_hip_pool_key_create`)."""
        base = self._parent if self._parent is not None else self
        return Pool(base.params, base.devices, _view_of=base)

    def close(self) -> None:
        if getattr(self, "_h", None):
            for ref in getattr(self, "_views", []):  # key views go before the pool they run on
                v = ref()
                if v is not None:
                    v.close()
            self._views = []
            self._lib.tfhe_hip_pool_destroy(self._h)
            self._h = None
            self._parent = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self) -> int:
        return int(self._lib.tfhe_hip_pool_size(self._h))

    def set_combining(self, max_count: int) -> None:
        """`Engine.set_combining` on every member (small concurrent calls go to the least loaded member's front end)."""
        for i in range(len(self)):
            Engine.from_pool(self, i).set_combining(max_count)

    def combine_stats(self) -> list:
        """`Engine.combine_stats` of every member."""
        return [Engine.from_pool(self, i).combine_stats() for i in range(len(self))]

    def _chk(self, rc: int) -> None:
        if rc != _capi.OK:
            msg = self._lib.tfhe_hip_pool_last_error(self._h)
            raise _capi.TfheHipError(rc, msg.decode() if msg else "")

    @property
    def key_transport(self) -> str:
        """"rccl" / "peer-copy": how the last cloud key reached the members (`tfhe_hip_pool_key_transport`)."""
        return self._lib.tfhe_hip_pool_key_transport(self._h).decode()

    def members_for(self, count: int) -> int:
        """Members a batch of `count` is spread over (`tfhe_hip_pool_members_for`): small batches use fewer."""
        return int(self._lib.tfhe_hip_pool_members_for(self._h, count))

    def shard(self, count: int, member: int) -> tuple:
        """[lo, hi) of the batch that `member` runs -- over the members the call actually uses (members_for),
        so (0, 0) for a member a small batch leaves idle."""
        lo, hi = C.c_size_t(0), C.c_size_t(0)
        self._lib.tfhe_hip_pool_shard(count, member, self.members_for(count), C.byref(lo), C.byref(hi))
        return int(lo.value), int(hi.value)

    # -- cloud key ------------------------------------------------------------
    def load_cloud_key(self, cloud_key) -> None:
        p = self.params
        bsk = np.ascontiguousarray(cloud_key.bootstrapping_key, dtype=np.float64)
        ksk, tv = _u32(cloud_key.key_switching_key), _u32(cloud_key.blind_rotate_testvec)
        if bsk.size != p.n * 2 * p.l * 2 * N or ksk.size != N * p.iks_t * p.base * (p.n + 1) or tv.size != 2 * N:
            raise ValueError("cloud key has the wrong size for these parameters")
        self._chk(self._lib.tfhe_hip_pool_load_cloud_key(self._h, _ptr(bsk), _ptr(ksk),
                                                         C.c_uint32(int(cloud_key.decomposition_offset)), _ptr(tv)))

    def gen_cloud_key(self, key_lv0, key_lv1, seed=None) -> None:
        """seed=None: generator keyed by the OS (tfhe_hip_pool_gen_cloud_key_secure); an integer: tests only."""
        p = self.params
        k0, k1 = _u32(key_lv0).reshape(-1), _u32(key_lv1).reshape(-1)
        if len(k0) != p.n or len(k1) != N:
            raise ValueError("secret key has the wrong size for these parameters")
        a0, a1 = C.c_double(p.alpha_lv0), C.c_double(p.alpha_lv1)
        if seed is None:
            self._chk(self._lib.tfhe_hip_pool_gen_cloud_key_secure(self._h, _ptr(k0), _ptr(k1), a0, a1))
        else:
            self._chk(self._lib.tfhe_hip_pool_gen_cloud_key(self._h, _ptr(k0), _ptr(k1), a0, a1,
                                                            C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF)))

    def export_cloud_key(self, member: int = 0):
        from .key import CloudKey

        p = self.params
        bsk = np.empty((p.n, 2 * p.l, 2, N), np.float64)
        ksk = np.empty((N, p.iks_t, p.base, p.n + 1), np.uint32)
        tv = np.empty((2, N), np.uint32)
        off = C.c_uint32(0)
        self._chk(self._lib.tfhe_hip_pool_export_cloud_key(self._h, member, _ptr(bsk), _ptr(ksk), C.byref(off), _ptr(tv)))
        return CloudKey(p, bsk, ksk, int(off.value), tv)

    # -- batched hot path, host arrays ------------------------------------------
    def _cts(self, a) -> np.ndarray:
        return _u32(a).reshape(-1, self.params.n + 1)

    def batch_gate(self, gate: int, a, b=None, out=None) -> np.ndarray:
        a = self._cts(a)
        bb = self._cts(b) if b is not None else None
        if bb is not None and bb.shape != a.shape:
            raise ValueError("operand batches differ in shape")
        out = _out_like(a, out)
        self._chk(self._lib.tfhe_hip_pool_batch_gate(self._h, int(gate), _ptr(a), _ptr(bb), _ptr(out), len(a)))
        return out

    def batch_gates_mixed(self, gates, a, b, keyswitch: bool = True) -> np.ndarray:
        a, b = self._cts(a), self._cts(b)
        g = np.ascontiguousarray(gates, dtype=np.uint8).reshape(-1)
        if len(g) != len(a) or b.shape != a.shape:
            raise ValueError("gates / operand batches differ in length")
        out = np.empty_like(a)
        fn = self._lib.tfhe_hip_pool_batch_gates_mixed if keyswitch else self._lib.tfhe_hip_pool_batch_gates_mixed_nks
        self._chk(fn(self._h, _ptr(g), _ptr(a), _ptr(b), _ptr(out), len(a)))
        return out

    def batch_tlwe_lincomb(self, ca: int, a, cb: int = 0, b=None, cconst: int = 0) -> np.ndarray:
        a = self._cts(a)
        bb = self._cts(b) if b is not None else None
        if (cb & 0xFFFFFFFF) and (bb is None or bb.shape != a.shape):
            raise ValueError("second operand missing or of a different shape")
        out = np.empty_like(a)
        self._chk(self._lib.tfhe_hip_pool_batch_tlwe_lincomb(
            self._h, ca & 0xFFFFFFFF, _ptr(a), cb & 0xFFFFFFFF, _ptr(bb), cconst & 0xFFFFFFFF, _ptr(out), len(a)))
        return out

    def batch_lincomb_bootstrap(self, ca: int, a, cb: int = 0, b=None, cconst: int = 0, testvec=None,
                                keyswitch: bool = True) -> np.ndarray:
        a = self._cts(a)
        bb = self._cts(b) if b is not None else None
        if (cb & 0xFFFFFFFF) and (bb is None or bb.shape != a.shape):
            raise ValueError("second operand missing or of a different shape")
        tv, per_ct = None, 0
        if testvec is not None:
            tv = _u32(testvec)
            per_ct = int(tv.ndim == 3)
            if tv.size != (len(a) if per_ct else 1) * 2 * N:
                raise ValueError("test vector must be [2][N], or [count][2][N] for per-ciphertext tables")
        out = np.empty_like(a)
        self._chk(self._lib.tfhe_hip_pool_batch_lincomb_bootstrap(
            self._h, ca & 0xFFFFFFFF, _ptr(a), cb & 0xFFFFFFFF, _ptr(bb), cconst & 0xFFFFFFFF, _ptr(tv), per_ct,
            int(keyswitch), _ptr(out), len(a)))
        return out

    def batch_bootstrap(self, cts, testvec=None, keyswitch: bool = True) -> np.ndarray:
        cts = self._cts(cts)
        out = np.empty_like(cts)
        per_ct, tv = 0, None
        if testvec is not None:
            tv = _u32(testvec)
            per_ct = int(tv.ndim == 3)
            if tv.size != (len(cts) if per_ct else 1) * 2 * N:
                raise ValueError("testvec must be [2][N] or [count][2][N]")
        self._chk(self._lib.tfhe_hip_pool_batch_bootstrap(self._h, _ptr(cts), _ptr(tv), per_ct, int(keyswitch), _ptr(out),
                                                          len(cts)))
        return out

    def batch_mux(self, a, b, c, naive: bool) -> np.ndarray:
        a, b, c = self._cts(a), self._cts(b), self._cts(c)
        if not (a.shape == b.shape == c.shape):
            raise ValueError("operand batches differ in shape")
        out = np.empty_like(a)
        self._chk(self._lib.tfhe_hip_pool_batch_mux(self._h, int(bool(naive)), _ptr(a), _ptr(b), _ptr(c), _ptr(out), len(a)))
        return out

    def batch_blind_rotate(self, cts, testvec=None) -> np.ndarray:
        cts = self._cts(cts)
        tv = _u32(testvec) if testvec is not None else None
        out = np.empty((len(cts), 2, N), np.uint32)
        self._chk(self._lib.tfhe_hip_pool_batch_blind_rotate(self._h, _ptr(cts), _ptr(tv), _ptr(out), len(cts)))
        return out

    # -- a batch resident on ONE member's GPU (torch CUDA tensors on member `home`'s device; enqueue only) ----------
    # Same signatures as Engine's *_dev methods plus `home` (default: self.home), so Circuit.run_dev,
    # circuit.mux_and_gates_dev and circuit.lut_add_u8_dev take a Pool wherever they take an Engine: shard 0 is
    # computed in place on the home GPU, the others travel by grouped RCCL send / receive (or peer copies) and come
    # back in input order (`tfhe_hip_pool_batch_*_dev`, include/tfhe_hip.h).
    @property
    def device(self) -> int:
        return self.devices[self.home]

    def _home(self, home) -> int:
        h = self.home if home is None else int(home)
        if not 0 <= h < len(self.devices):
            raise ValueError("no such pool member")
        return h

    def _dev_batch(self, home: int, *tensors, width=None) -> int:
        width = self.params.n + 1 if width is None else width
        first = tensors[0]
        for t in tensors:
            if t is None:
                continue
            if t.dim() != 2 or t.shape[1] != width or t.shape[0] != first.shape[0]:
                raise ValueError(f"device tensors must all be [count][{width}]")
            if t.device.index != self.devices[home]:
                raise ValueError(f"device tensor lives on {t.device}, the home member on cuda:{self.devices[home]}")
        return first.shape[0]

    def _tp(self, home: int, t):
        p = _tptr(t)
        if t is not None and t.device.index != self.devices[home]:
            raise ValueError(f"device tensor lives on {t.device}, the home member on cuda:{self.devices[home]}")
        return p

    def _stream_ptr(self, home: int, stream):
        """The HOME member's stream: torch's current stream OF THAT DEVICE (the torch-current device may be another
        GPU: a handle from there would be an invalid resource on the home GPU after the scatter was enqueued), or the
        caller's stream, which must live on the home member's device."""
        import torch

        dev = self.devices[home]
        if stream is None:
            stream = torch.cuda.current_stream(dev)
        elif getattr(stream, "device", None) is not None and stream.device.index != dev:
            raise ValueError(f"stream lives on {stream.device}, the home member on cuda:{dev}")
        return C.c_void_p(stream.cuda_stream or 1)

    def batch_gate_dev(self, gate: int, a, b, out, stream=None, home=None) -> None:
        h = self._home(home)
        count = self._dev_batch(h, a, b, out)
        self._chk(self._lib.tfhe_hip_pool_batch_gate_dev(self._h, h, int(gate), self._tp(h, a), self._tp(h, b), self._tp(h, out),
                                                         count, self._stream_ptr(h, stream)))

    def batch_gates_mixed_dev(self, gates, a, b, out, stream=None, keyswitch: bool = True, home=None) -> None:
        h = self._home(home)
        if not gates.is_cuda or gates.element_size() != 1 or not gates.is_contiguous() or gates.device.index != self.devices[h]:
            raise ValueError("gates must be a contiguous uint8 CUDA tensor on the home member's GPU")
        count = self._dev_batch(h, a, b, out)
        if gates.numel() != count:
            raise ValueError("one gate code per ciphertext")
        fn = self._lib.tfhe_hip_pool_batch_gates_mixed_dev if keyswitch else self._lib.tfhe_hip_pool_batch_gates_mixed_nks_dev
        self._chk(fn(self._h, h, C.c_void_p(gates.data_ptr()), self._tp(h, a), self._tp(h, b), self._tp(h, out), count,
                     self._stream_ptr(h, stream)))

    def batch_bootstrap_dev(self, cts, out, testvec=None, per_ct: bool = False, keyswitch: bool = True, stream=None,
                            home=None) -> None:
        h = self._home(home)
        count = self._dev_batch(h, cts, out)
        if testvec is not None and testvec.numel() != (count if per_ct else 1) * 2 * N:
            raise ValueError("test vector must be [2][N], or [count][2][N] with per_ct")
        self._chk(self._lib.tfhe_hip_pool_batch_bootstrap_dev(self._h, h, self._tp(h, cts), self._tp(h, testvec), int(per_ct),
                                                              int(keyswitch), self._tp(h, out), count, self._stream_ptr(h, stream)))

    def batch_tlwe_lincomb_dev(self, ca: int, a, cb: int, b, cconst: int, out, stream=None, home=None) -> None:
        h = self._home(home)
        count = self._dev_batch(h, a, b, out)
        self._chk(self._lib.tfhe_hip_pool_batch_tlwe_lincomb_dev(
            self._h, h, ca & 0xFFFFFFFF, self._tp(h, a), cb & 0xFFFFFFFF, self._tp(h, b), cconst & 0xFFFFFFFF, self._tp(h, out),
            count, self._stream_ptr(h, stream)))

    def batch_lincomb_bootstrap_dev(self, ca: int, a, cb: int, b, cconst: int, out, testvec=None, per_ct: bool = False,
                                    keyswitch: bool = True, stream=None, home=None) -> None:
        h = self._home(home)
        count = self._dev_batch(h, a, b, out)
        if testvec is not None and testvec.numel() != (count if per_ct else 1) * 2 * N:
            raise ValueError("test vector must be [2][N], or [count][2][N] with per_ct")
        self._chk(self._lib.tfhe_hip_pool_batch_lincomb_bootstrap_dev(
            self._h, h, ca & 0xFFFFFFFF, self._tp(h, a), cb & 0xFFFFFFFF, self._tp(h, b), cconst & 0xFFFFFFFF,
            self._tp(h, testvec), int(per_ct), int(keyswitch), self._tp(h, out), count, self._stream_ptr(h, stream)))

    def batch_mux_dev(self, a, b, c, out, naive: bool, stream=None, home=None) -> None:
        h = self._home(home)
        count = self._dev_batch(h, a, b, c, out)
        self._chk(self._lib.tfhe_hip_pool_batch_mux_dev(self._h, h, int(naive), self._tp(h, a), self._tp(h, b), self._tp(h, c),
                                                        self._tp(h, out), count, self._stream_ptr(h, stream)))

    def batch_blind_rotate_dev(self, cts, out_trlwe, testvec=None, stream=None, home=None) -> None:
        h = self._home(home)
        count = self._dev_batch(h, cts)
        if out_trlwe.numel() != count * 2 * N or (testvec is not None and testvec.numel() != 2 * N):
            raise ValueError("out_trlwe must be [count][2][N], testvec [2][N]")
        self._chk(self._lib.tfhe_hip_pool_batch_blind_rotate_dev(self._h, h, self._tp(h, cts), self._tp(h, testvec),
                                                                 self._tp(h, out_trlwe), count, self._stream_ptr(h, stream)))

    def synchronize(self) -> None:
        """Drain every member's own stream (`tfhe_hip_pool_synchronize`); the home stream is the caller's."""
        self._chk(self._lib.tfhe_hip_pool_synchronize(self._h))

    @property
    def data_transport(self) -> str:
        """"rccl" / "peer-copy" / "none": how the last *_dev call moved its shards."""
        return self._lib.tfhe_hip_pool_data_transport(self._h).decode()

    def set_profiling(self, enabled: bool) -> None:
        self._chk(self._lib.tfhe_hip_pool_set_profiling(self._h, int(enabled)))

    def transfer_times(self) -> dict:
        tt = _capi.PoolTransferTimes()
        self._chk(self._lib.tfhe_hip_pool_get_transfer_times(self._h, C.byref(tt)))
        return {k: getattr(tt, k) for k, _ in _capi.PoolTransferTimes._fields_}
