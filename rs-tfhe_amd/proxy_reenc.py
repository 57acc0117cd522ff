"""LWE proxy re-encryption (the reference's feature `proxy-reenc`, src/proxy_reenc.rs).

    PublicKeyLv0::new / new_with_params / encrypt_f64 / encrypt_bool       src/proxy_reenc.rs:95-222
    ProxyReencryptionKey::new_asymmetric[_with_params]                     src/proxy_reenc.rs:271-330
    ProxyReencryptionKey::new_symmetric[_with_params]                      src/proxy_reenc.rs:362-425
    reencrypt_tlwe_lv0                                                     src/proxy_reenc.rs:468-510

Key generation is client-side integer work (numpy, batched; it needs the delegator's secret key).  The re-encryption
itself -- the proxy's job, one digit-lookup walk over n * t rows of n + 1 words per ciphertext, i.e. the identity key
switch with a source of n coefficients -- runs on the GPU through the key-switch kernels
(`tfhe_hip_load_reenc_key` / `tfhe_hip_batch_reencrypt`, include/tfhe_hip.h); there is no CPU path here.

Randomness follows client.py: `seed=None` draws from the operating system; an integer seed is reproducible and for
tests only.
"""
from __future__ import annotations

import numpy as np

from .client import SecretKey, _rng, f64_to_torus
from .params import SecurityParams


def _key_lv0(key) -> np.ndarray:
    return np.ascontiguousarray(key.key_lv0 if isinstance(key, SecretKey) else key, dtype=np.uint32)


class PublicKeyLv0:
    """proxy_reenc.rs:95-99: encryptions of zero under the secret key, [size][n+1]."""

    def __init__(self, params: SecurityParams, encryptions):
        self.params = params
        self.encryptions = np.ascontiguousarray(encryptions, dtype=np.uint32).reshape(-1, params.n + 1)

    @classmethod
    def new(cls, secret_key: SecretKey, seed=None) -> "PublicKeyLv0":
        """proxy_reenc.rs:125-131: 2n encryptions of zero at the level-0 noise."""
        p = secret_key.params
        return cls.new_with_params(secret_key, 2 * p.n, p.alpha_lv0, seed)

    @classmethod
    def new_with_params(cls, secret_key: SecretKey, size: int, alpha: float, seed=None) -> "PublicKeyLv0":
        """proxy_reenc.rs:144-153."""
        return cls(secret_key.params, secret_key.encrypt_f64(np.zeros(int(size)), seed, alpha))

    def encrypt_f64(self, plaintext, alpha: float, seed=None) -> np.ndarray:
        """proxy_reenc.rs:168-200, batched over `plaintext`: every encryption of zero joins with probability 1/2, added
        or subtracted with probability 1/2 each; then f64_to_torus(plaintext) and fresh noise N(0, alpha) on b."""
        g = _rng(seed)
        pt = np.atleast_1d(np.asarray(plaintext, dtype=np.float64))
        size, w = self.encryptions.shape
        enc = self.encryptions.astype(np.float64)  # |sum| <= size * 2^32 < 2^53: the f64 product below is exact
        out = np.empty((len(pt), w), np.uint32)
        for lo in range(0, len(pt), 2048):
            m = min(2048, len(pt) - lo)
            take = g.integers(0, 2, (m, size), dtype=np.uint32).astype(np.float64)
            sign = 1.0 - 2.0 * g.integers(0, 2, (m, size), dtype=np.uint32).astype(np.float64)
            acc = (take * sign) @ enc
            out[lo:lo + m] = np.mod(acc, 4294967296.0).astype(np.uint64).astype(np.uint32)
        noise = f64_to_torus(g.normal(0.0, alpha, len(pt))) if alpha > 0 else np.zeros(len(pt), np.uint32)
        out[:, -1] += f64_to_torus(pt) + noise
        return out

    def encrypt_bool(self, bits, alpha: float, seed=None) -> np.ndarray:
        """proxy_reenc.rs:212-215."""
        bits = np.atleast_1d(np.asarray(bits)).astype(bool)
        return self.encrypt_f64(np.where(bits, 0.125, -0.125), alpha, seed)


class ProxyReencryptionKey:
    """proxy_reenc.rs:224-233: key_encryptions [n][t][base][n+1] (index base*t*i + base*j + k; the k = 0 entries stay
    zero, :311-313), base, t.  `reencrypt` keeps the key resident on the GPU in a key view of the shared context."""

    def __init__(self, params: SecurityParams, key_encryptions, basebit: int, t: int):
        if params.n > 1024:  # (the key-switch kernels' row count is N = 1024: said here, not at the first reencrypt)
            raise ValueError(f"proxy re-encryption on the GPU needs n <= 1024 ({params.name}: n = {params.n})")
        self.params = params
        self.basebit, self.t, self.base = int(basebit), int(t), 1 << int(basebit)
        self.key_encryptions = np.ascontiguousarray(key_encryptions, dtype=np.uint32).reshape(
            params.n * self.t * self.base, params.n + 1)
        self._view = None

    # the plaintexts both constructors encrypt: k * key_from[i] / 2^((j+1) basebit), k = 1 .. base-1 (:316, :414)
    @staticmethod
    def _plaintexts(key_from: np.ndarray, basebit: int, t: int) -> np.ndarray:
        base = 1 << basebit
        k = np.arange(base, dtype=np.uint32)[None, None, :]
        j = np.arange(t)[None, :, None]
        val = (k * key_from[:, None, None]).astype(np.uint32).astype(np.float64)
        return val / (1 << ((j + 1) * basebit)).astype(np.float64)  # [n][t][base]

    @classmethod
    def new_symmetric(cls, key_from, key_to: SecretKey, seed=None) -> "ProxyReencryptionKey":
        """proxy_reenc.rs:362-370: the set's key-switch noise, basebit and t."""
        p = key_to.params
        return cls.new_symmetric_with_params(key_from, key_to, p.alpha_lv0, p.basebit, p.iks_t, seed)

    @classmethod
    def new_symmetric_with_params(cls, key_from, key_to: SecretKey, alpha: float, basebit: int, t: int,
                                  seed=None) -> "ProxyReencryptionKey":
        """proxy_reenc.rs:389-425: TLWELv0::encrypt_f64(p, alpha, key_to) per (i, j, k != 0)."""
        p = key_to.params
        pts = cls._plaintexts(_key_lv0(key_from), basebit, t)
        enc = key_to.encrypt_f64(pts.reshape(-1), seed, alpha).reshape(p.n, t, 1 << basebit, p.n + 1)
        enc[:, :, 0, :] = 0
        return cls(p, enc, basebit, t)

    @classmethod
    def new_asymmetric(cls, key_from, public_key_to: PublicKeyLv0, seed=None) -> "ProxyReencryptionKey":
        """proxy_reenc.rs:271-279."""
        p = public_key_to.params
        return cls.new_asymmetric_with_params(key_from, public_key_to, p.alpha_lv0, p.basebit, p.iks_t, seed)

    @classmethod
    def new_asymmetric_with_params(cls, key_from, public_key_to: PublicKeyLv0, alpha: float, basebit: int, t: int,
                                   seed=None) -> "ProxyReencryptionKey":
        """proxy_reenc.rs:294-330: public_key_to.encrypt_f64(p, alpha) per (i, j, k != 0)."""
        p = public_key_to.params
        pts = cls._plaintexts(_key_lv0(key_from), basebit, t)
        enc = public_key_to.encrypt_f64(pts.reshape(-1), alpha, seed).reshape(p.n, t, 1 << basebit, p.n + 1)
        enc[:, :, 0, :] = 0
        return cls(p, enc, basebit, t)

    # ---- the proxy's side: on the GPU -------------------------------------------------------------------------
    def _engine_params(self) -> SecurityParams:
        """The context's parameter set: the ciphertexts' set with this key's (basebit, t) (custom `_with_params` keys)."""
        p = self.params
        if (p.basebit, p.iks_t) == (self.basebit, self.t):
            return p
        import dataclasses

        return dataclasses.replace(p, name=f"{p.name}+reenc(basebit={self.basebit},t={self.t})", basebit=self.basebit, iks_t=self.t)

    def view(self, device: int = 0):
        """The key view that holds this key (created and loaded on first use; `close()` frees its 0.1 GB)."""
        if self._view is None or self._view[0] != device:
            from .bootstrap import engine_for

            self.close()
            v = engine_for(self._engine_params(), device).new_key_view()
            v.load_reenc_key(self.key_encryptions)
            self._view = (device, v)
        return self._view[1]

    def close(self) -> None:
        """Free the key's device memory (also on garbage collection, and at the end of a `with` block)."""
        if getattr(self, "_view", None) is not None:
            self._view[1].close()
            self._view = None

    def __enter__(self) -> "ProxyReencryptionKey":
        return self

    def __exit__(self, *exc) -> None:
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reencrypt(self, cts, device: int = 0) -> np.ndarray:
        """reencrypt_tlwe_lv0 (proxy_reenc.rs:468-510) over [count][n+1] (or one [n+1]) ciphertexts."""
        arr = np.ascontiguousarray(cts, dtype=np.uint32)
        out = self.view(device).batch_reencrypt(arr.reshape(-1, self.params.n + 1))
        return out.reshape(arr.shape)


def reencrypt_tlwe_lv0(ct_from, reenc_key: ProxyReencryptionKey, device: int = 0) -> np.ndarray:
    """proxy_reenc.rs:468: the free function of the reference; accepts a batch as well."""
    return reenc_key.reencrypt(ct_from, device)
