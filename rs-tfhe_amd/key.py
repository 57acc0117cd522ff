"""CloudKey: the evaluation key bundle of the reference (src/key.rs:51-56).

This class carries the four fields the hot path borrows, in the flat layouts of
include/tfhe_hip.h.  `CloudKey.new(secret_key)` (src/key.rs:59-66) generates them on
the GPU (client.SecretKey.cloud_key -> tfhe_hip_gen_cloud_key).
"""
from __future__ import annotations

import numpy as np

from .params import N, SecurityParams, f64_to_torus, gen_decomposition_offset


def gen_testvec() -> np.ndarray:
    """src/key.rs:91-100: a = 0, b = f64_to_torus(0.125)."""
    tv = np.zeros((2, N), np.uint32)
    tv[1, :] = f64_to_torus(0.125)
    return tv


class CloudKey:
    @classmethod
    def new(cls, secret_key, seed=None, device: int = 0) -> "CloudKey":
        """CloudKey::new(&secret_key), src/key.rs:59-66.  seed=None draws the generator key from the OS;
        an integer seed is for reproducible tests only (see client.SecretKey.cloud_key)."""
        return secret_key.cloud_key(seed, device)

    def __init__(self, params: SecurityParams, bootstrapping_key, key_switching_key,
                 decomposition_offset=None, blind_rotate_testvec=None):
        self.params = params
        self.decomposition_offset = (
            gen_decomposition_offset(params) if decomposition_offset is None else int(decomposition_offset)
        )
        self.blind_rotate_testvec = gen_testvec() if blind_rotate_testvec is None else np.ascontiguousarray(
            blind_rotate_testvec, dtype=np.uint32
        ).reshape(2, N)
        self.bootstrapping_key = np.ascontiguousarray(bootstrapping_key, dtype=np.float64).reshape(
            params.n, 2 * params.l, 2, N
        )
        self.key_switching_key = np.ascontiguousarray(key_switching_key, dtype=np.uint32).reshape(
            N, params.iks_t, params.base, params.n + 1
        )
