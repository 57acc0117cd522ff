"""Run-time parameter sets: the data of src/params.rs:91-404 of the reference.

The reference binds every type to SECURITY_128_BIT at compile time
(src/params.rs:426-465); the engine takes (n, l, bgbit, basebit, t) at run
time so the other sets are reachable.  N = 1024 in every set.
"""
from __future__ import annotations

from dataclasses import dataclass

N = 1024
NBIT = 10
TORUS_SIZE = 32


@dataclass(frozen=True)
class SecurityParams:
    name: str
    security_bits: int
    n: int  # tlwe_lv0.n
    l: int  # trgsw_lv1.l
    bgbit: int  # trgsw_lv1.bgbit
    basebit: int  # trgsw_lv1.basebit
    iks_t: int  # trgsw_lv1.iks_t
    alpha_lv0: float  # tlwe_lv0.alpha
    alpha_lv1: float  # tlwe_lv1.alpha

    @property
    def t(self) -> int:
        return self.iks_t

    @property
    def bg(self) -> int:
        return 1 << self.bgbit

    @property
    def base(self) -> int:
        return 1 << self.basebit

    # sizes used by the roofline accounting
    @property
    def tlwe_lv0_bytes(self) -> int:
        return (self.n + 1) * 4

    @property
    def bsk_bytes(self) -> int:
        return self.n * 2 * self.l * 2 * N * 8

    @property
    def ksk_bytes(self) -> int:
        return self.base * self.iks_t * N * (self.n + 1) * 4

    @property
    def ksk_touched_bytes(self) -> int:
        """Expected KSK bytes one key switch reads: N*t*(1-1/base)*(n+1)*4."""
        return int(N * self.iks_t * (1.0 - 1.0 / self.base) * (self.n + 1) * 4)

    def algorithmic_bytes_per_bootstrap(self, n_inputs: int = 2) -> int:
        """BSK once + touched KSK rows + ciphertext I/O."""
        return self.bsk_bytes + self.ksk_touched_bytes + (n_inputs + 1) * self.tlwe_lv0_bytes


# src/params.rs:91-116
SECURITY_80_BIT = SecurityParams("SECURITY_80_BIT", 80, 550, 3, 6, 2, 7, 5.0e-5, 3.73e-8)
# src/params.rs:119-144
SECURITY_110_BIT = SecurityParams("SECURITY_110_BIT", 110, 630, 3, 6, 2, 8, 3.0517578125e-05, 2.9802322387695313e-8)
# src/params.rs:379-404
SECURITY_128_BIT = SecurityParams("SECURITY_128_BIT", 128, 700, 3, 6, 2, 9, 2.0e-5, 2.0e-8)
# src/params.rs:148-173
SECURITY_UINT1 = SecurityParams("SECURITY_UINT1", 1, 700, 2, 10, 2, 8, 2.0e-05, 2.0e-08)
# src/params.rs:177-202
SECURITY_UINT2 = SecurityParams("SECURITY_UINT2", 2, 687, 1, 18, 4, 3, 0.00002120846893069972, 0.0000000000023184122752704995)
# src/params.rs:206-231
SECURITY_UINT3 = SecurityParams("SECURITY_UINT3", 3, 820, 1, 23, 6, 2, 0.0000025167616095979554, 2.220446049250313e-16)
# src/params.rs:235-260
SECURITY_UINT4 = SecurityParams("SECURITY_UINT4", 4, 820, 1, 22, 5, 3, 0.0000025167616095979554, 2.220446049250313e-16)
# src/params.rs:264-289
SECURITY_UINT5 = SecurityParams("SECURITY_UINT5", 5, 1071, 1, 22, 6, 3, 7.08822676541043e-8, 2.2204460492503131e-17)
# src/params.rs:293-318 (same shapes as UINT5: message modulus 64 is a property of the encoder, not of the kernels)
SECURITY_UINT6 = SecurityParams("SECURITY_UINT6", 6, 1071, 1, 22, 6, 3, 7.08822676541043e-8, 2.2204460492503131e-17)
# src/params.rs:322-347
SECURITY_UINT7 = SecurityParams("SECURITY_UINT7", 7, 1160, 1, 22, 7, 3, 1.9662200074984027e-8, 2.2204460492503131e-17)
# src/params.rs:351-376 (same shapes as UINT7)
SECURITY_UINT8 = SecurityParams("SECURITY_UINT8", 8, 1160, 1, 22, 7, 3, 1.9662200074984027e-8, 2.2204460492503131e-17)

DEFAULT_SECURITY = SECURITY_128_BIT  # src/params.rs:411

PARAM_SETS = {
    p.name: p
    for p in (
        SECURITY_80_BIT,
        SECURITY_110_BIT,
        SECURITY_128_BIT,
        SECURITY_UINT1,
        SECURITY_UINT2,
        SECURITY_UINT3,
        SECURITY_UINT4,
        SECURITY_UINT5,
        SECURITY_UINT6,
        SECURITY_UINT7,
        SECURITY_UINT8,
    )
}


def f64_to_torus(d: float) -> int:
    """src/utils.rs:9-12: ((d % 1.0) * 2^32) as i64 as u32."""
    import math

    torus = math.fmod(d, 1.0) * 4294967296.0
    return int(torus) & 0xFFFFFFFF


def torus_to_f64(t: int) -> float:
    """src/utils.rs:14-16"""
    return float(t) / 4294967296.0


def gen_decomposition_offset(p: SecurityParams) -> int:
    """src/key.rs:78-89"""
    off = 0
    for i in range(p.l):
        off = (off + (p.bg // 2) * (1 << (TORUS_SIZE - (i + 1) * p.bgbit))) & 0xFFFFFFFF
    return off
