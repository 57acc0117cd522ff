"""Lookup-table construction for programmable bootstrapping (host side, integer).

Mirrors src/lut/encoder.rs (Encoder), src/lut/generator.rs (Generator) and
src/lut/lookup_table.rs (LookupTable) of the reference.
"""
from __future__ import annotations

import numpy as np

from .params import N, f64_to_torus, torus_to_f64


def div_round(a: int, b: int) -> int:
    """src/lut/generator.rs:264-266"""
    return (a + b // 2) // b


class Encoder:
    """src/lut/encoder.rs:13-115"""

    def __init__(self, message_modulus: int, scale: float | None = None):
        self.message_modulus = int(message_modulus)
        self.scale = 1.0 / (2.0 * message_modulus) if scale is None else float(scale)

    @classmethod
    def with_scale(cls, message_modulus: int, scale: float) -> "Encoder":
        return cls(message_modulus, scale)

    def encode(self, message: int) -> int:  # encoder.rs:66-73
        message = int(message) % self.message_modulus
        return f64_to_torus(message * self.scale)

    def encode_with_scale(self, message: int, scale: float) -> int:  # encoder.rs:83-87
        return f64_to_torus((int(message) % self.message_modulus) * scale)

    def decode(self, value: int) -> int:  # encoder.rs:96-105
        f = torus_to_f64(int(value))
        return int(f / self.scale + 0.5) % self.message_modulus

    def decode_bool(self, value: int) -> bool:  # encoder.rs:113-115
        return self.decode(value) != 0


class LookupTable:
    """src/lut/lookup_table.rs:16-19: a TRLWE whose b polynomial encodes the function."""

    def __init__(self, poly=None):
        self.poly = np.zeros((2, N), np.uint32) if poly is None else np.ascontiguousarray(poly, np.uint32).reshape(2, N)

    @classmethod
    def from_poly(cls, poly) -> "LookupTable":
        return cls(poly)

    def copy_from(self, other: "LookupTable") -> None:
        self.poly[...] = other.poly

    def clear(self) -> None:
        self.poly[...] = 0

    def is_empty(self) -> bool:
        return not self.poly.any()


class Generator:
    """src/lut/generator.rs:15-259"""

    def __init__(self, message_modulus: int, scale: float | None = None):
        self.encoder = Encoder(message_modulus, scale)
        self.poly_degree = N
        self.lookup_table_size = N

    @classmethod
    def with_scale(cls, message_modulus: int, scale: float) -> "Generator":
        return cls(message_modulus, scale)

    @property
    def message_modulus(self) -> int:
        return self.encoder.message_modulus

    def _assemble(self, values) -> LookupTable:
        """generator.rs:89-137 given the per-message torus values."""
        m = self.message_modulus
        size = self.lookup_table_size
        raw = np.zeros(size, np.uint32)
        for x in range(m):
            start = div_round(x * size, m)
            end = min(div_round((x + 1) * size, m), size)
            raw[start:end] = values[x]
        offset = div_round(size, 2 * m)
        rot = raw[(np.arange(size) + offset) % size].copy()
        if offset:
            rot[size - offset:] = (0 - rot[size - offset:].astype(np.int64)).astype(np.uint32)  # wrapping_neg
        lut = LookupTable()
        lut.poly[0, :] = 0
        lut.poly[1, :] = rot
        return lut

    def generate_lookup_table(self, f) -> LookupTable:  # generator.rs:66-73
        m = self.message_modulus
        return self._assemble([self.encoder.encode(f(x)) for x in range(m)])

    def generate_lookup_table_assign(self, f, lut_out: LookupTable) -> None:  # generator.rs:89-137: into an existing table
        lut_out.copy_from(self.generate_lookup_table(f))

    def generate_lookup_table_full(self, f) -> LookupTable:  # generator.rs:146-153
        m = self.message_modulus
        return self._assemble([int(f(x)) & 0xFFFFFFFF for x in range(m)])

    def generate_lookup_table_full_assign(self, f, lut_out: LookupTable) -> None:  # generator.rs:160-203
        lut_out.copy_from(self.generate_lookup_table_full(f))

    def generate_lookup_table_custom(self, f, message_modulus: int, scale: float) -> LookupTable:  # :203-222
        return Generator(message_modulus, scale).generate_lookup_table(f)

    def mod_switch(self, x: int) -> int:  # generator.rs:232-235
        scaled = float(x) / float(0xFFFFFFFF) * float(self.lookup_table_size)
        r = int(np.floor(abs(scaled) + 0.5))  # f64::round, half away from zero
        return r % self.lookup_table_size
