"""Client side of the path: secret keys, TLWE encryption and decryption.

Host-side integer work (numpy, batched over ciphertexts); nothing here touches
the GPU except `cloud_key()`, which runs the key generation kernels.  Mirrors

  key::SecretKey::new                      src/key.rs:21-48
  TLWELv0::encrypt_f64 / encrypt_bool      src/tlwe.rs:37-58
  TLWELv0::decrypt_bool                    src/tlwe.rs:60-68
  TLWELv0::encrypt_lwe_message             src/tlwe.rs:84-98
  TLWELv0::decrypt_lwe_message             src/tlwe.rs:111-126
  utils::f64_to_torus / gaussian_f64       src/utils.rs:9-38
  CloudKey::new(&secret_key)               src/key.rs:59-66

The reference draws from `thread_rng` (an OS-seeded ChaCha CSPRNG).  Here `seed=None` -- the
default everywhere -- draws from the operating system's CSPRNG (`os.urandom`); passing an integer
seed or a numpy Generator selects numpy's PCG64 instead, which is reproducible and NOT
cryptographic: tests and benchmarks only.  A guessable generator behind `SecretKey.new`,
`encrypt_*` or `cloud_key` gives the secret key away.
"""
from __future__ import annotations

import os

import numpy as np

from .params import N, SecurityParams


def f64_to_torus(d) -> np.ndarray:
    """utils.rs:9-12, element-wise: ((d % 1.0) * 2^32) as i64 as u32 (fmod keeps the sign of d)."""
    t = np.fmod(np.asarray(d, dtype=np.float64), 1.0) * 4294967296.0
    return t.astype(np.int64).astype(np.uint32)


def torus_to_f64(t) -> np.ndarray:
    """utils.rs:14-16."""
    return np.asarray(t, dtype=np.uint32).astype(np.float64) / 4294967296.0


class OsRng:
    """The two draws this module needs, fed by os.urandom (getrandom(2)): the stand-in for the
    reference's OS-seeded thread_rng."""

    def integers(self, low, high, size, dtype=np.uint64):
        span = int(high) - int(low)
        shape = (size,) if np.isscalar(size) else tuple(size)
        count = int(np.prod(shape))
        if span == 2:
            raw = np.frombuffer(os.urandom((count + 7) // 8), np.uint8)
            vals = np.unpackbits(raw)[:count].astype(np.uint64)
        elif span == 1 << 32:
            vals = np.frombuffer(os.urandom(4 * count), np.uint32).astype(np.uint64)
        else:
            raise ValueError("OsRng.integers serves bits and 32-bit words")
        return (vals + np.uint64(int(low))).astype(dtype).reshape(shape)

    def normal(self, mu, sigma, size):
        """Box-Muller over 53-bit uniforms (as the GPU key generator does, keygen.hpp gauss2)."""
        count = int(size)
        u = np.frombuffer(os.urandom(16 * ((count + 1) // 2)), np.uint64).reshape(-1, 2) >> np.uint64(11)
        u1 = (u[:, 0].astype(np.float64) + 1.0) * (1.0 / 9007199254740992.0)  # (0, 1]
        u2 = u[:, 1].astype(np.float64) * (1.0 / 9007199254740992.0)          # [0, 1)
        r = np.sqrt(-2.0 * np.log(u1)) * sigma
        return (mu + np.concatenate([r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)]))[:count]


def _rng(seed):
    """None -> OsRng (cryptographic); int / numpy Generator -> PCG64 (reproducible, tests only)."""
    if seed is None:
        return OsRng()
    return seed if isinstance(seed, (np.random.Generator, OsRng)) else np.random.default_rng(seed)


class SecretKey:
    """key_lv0 in {0,1}^n (the TLWE key ciphertexts live under), key_lv1 in {0,1}^N (the ring key)."""

    def __init__(self, params: SecurityParams, key_lv0, key_lv1):
        self.params = params
        self.key_lv0 = np.ascontiguousarray(key_lv0, dtype=np.uint32).reshape(params.n)
        self.key_lv1 = np.ascontiguousarray(key_lv1, dtype=np.uint32).reshape(N)
        if self.key_lv0.max(initial=0) > 1 or self.key_lv1.max(initial=0) > 1:
            raise ValueError("secret keys are binary")

    @classmethod
    def new(cls, params: SecurityParams, seed=None) -> "SecretKey":
        """key.rs:33-48: uniform bits."""
        g = _rng(seed)
        return cls(params, g.integers(0, 2, params.n, dtype=np.uint32), g.integers(0, 2, N, dtype=np.uint32))

    # ---- TLWE level 0 -----------------------------------------------------------------------
    def _inner(self, cts: np.ndarray) -> np.ndarray:
        # sum of the mask words the key selects, wrapping in u32
        return (cts[:, :-1] * self.key_lv0[None, :]).sum(axis=1, dtype=np.uint32)

    def _cts(self, cts) -> np.ndarray:
        return np.ascontiguousarray(cts, dtype=np.uint32).reshape(-1, self.params.n + 1)

    def encrypt_f64(self, p, seed=None, alpha: float | None = None) -> np.ndarray:
        """tlwe.rs:37-53: a uniform, b = <a, s> + f64_to_torus(p) + f64_to_torus(N(0, alpha)).
        p: scalar or [count]; returns [count][n+1] u32."""
        g = _rng(seed)
        p = np.atleast_1d(np.asarray(p, dtype=np.float64))
        alpha = self.params.alpha_lv0 if alpha is None else alpha
        out = np.empty((len(p), self.params.n + 1), np.uint32)
        out[:, :-1] = g.integers(0, 1 << 32, (len(p), self.params.n), dtype=np.uint64).astype(np.uint32)
        noise = f64_to_torus(g.normal(0.0, alpha, len(p))) if alpha > 0 else np.zeros(len(p), np.uint32)
        out[:, -1] = self._inner(out) + f64_to_torus(p) + noise
        return out

    def encrypt_bool(self, bits, seed=None, alpha: float | None = None) -> np.ndarray:
        """tlwe.rs:55-58: true -> +1/8, false -> -1/8."""
        bits = np.atleast_1d(np.asarray(bits)).astype(bool)
        return self.encrypt_f64(np.where(bits, 0.125, -0.125), seed, alpha)

    def phase(self, cts) -> np.ndarray:
        """b - <a, s> (u32)."""
        cts = self._cts(cts)
        return cts[:, -1] - self._inner(cts)

    def decrypt_bool(self, cts) -> np.ndarray:
        """tlwe.rs:60-68: the phase read as i32 is non-negative."""
        return self.phase(cts).view(np.int32) >= 0

    def encrypt_lwe_message(self, msgs, message_modulus: int, seed=None, alpha: float | None = None) -> np.ndarray:
        """tlwe.rs:84-98: (msg mod m) / (2m)."""
        m = int(message_modulus)
        msgs = np.atleast_1d(np.asarray(msgs)).astype(np.int64) % m
        return self.encrypt_f64(msgs.astype(np.float64) * (1.0 / (2.0 * m)), seed, alpha)

    def decrypt_lwe_message(self, cts, message_modulus: int) -> np.ndarray:
        """tlwe.rs:111-126: ((phase / 2^32) / scale + 0.5) as usize % m, scale = 1/(2m)."""
        m = int(message_modulus)
        scale = 1.0 / (2.0 * m)
        return (torus_to_f64(self.phase(cts)) / scale + 0.5).astype(np.int64) % m

    # ---- evaluation key ----------------------------------------------------------------------
    def cloud_key(self, seed=None, device: int = 0):
        """CloudKey::new(&secret_key) (key.rs:59-66): generated on the GPU in a fresh key view of the shared context
        for `device` (which stays resident as this key's view: first use uploads nothing), and returned in the
        reference layouts.  seed=None: the generator key comes from the OS (`tfhe_hip_gen_cloud_key_secure`); an
        integer seed gives a reproducible, guessable key (tests only)."""
        from .bootstrap import adopt_view, engine_for

        view = engine_for(self.params, device).new_key_view()
        view.gen_cloud_key(self.key_lv0, self.key_lv1, seed)
        ck = view.export_cloud_key()
        adopt_view(ck, view)
        return ck
