"""Bootstrap strategies behind the reference's `trait Bootstrap`
(src/bootstrap/mod.rs:23-38), backed by the HIP engine.

    Bootstrap.bootstrap(ctxt, cloud_key) -> Ciphertext
    Bootstrap.bootstrap_without_key_switch(ctxt, cloud_key) -> Ciphertext
    Bootstrap.name() -> str

A Ciphertext is a numpy uint32 array of n+1 words (TLWELv0, src/tlwe.rs:12-14);
every method also accepts a [count][n+1] batch, which is how the GPU is meant
to be fed.
"""
from __future__ import annotations

import abc

import numpy as np

from .engine import COPY, Engine
from .lut import Generator, LookupTable
from .params import DEFAULT_SECURITY, SecurityParams

_engines: dict = {}


def engine_for(params: SecurityParams, device: int = 0) -> Engine:
    """One Engine (C-ABI context) per (parameter set, device) in this process."""
    key = (params, device)
    eng = _engines.get(key)
    if eng is None:
        eng = _engines[key] = Engine(params, device)
    return eng


def _params_of(cloud_key) -> SecurityParams:
    return getattr(cloud_key, "params", DEFAULT_SECURITY)


class Bootstrap(abc.ABC):
    """src/bootstrap/mod.rs:23-38"""

    @abc.abstractmethod
    def bootstrap(self, ctxt, cloud_key): ...

    @abc.abstractmethod
    def bootstrap_without_key_switch(self, ctxt, cloud_key): ...

    @abc.abstractmethod
    def name(self) -> str: ...


class HipBootstrap(Bootstrap):
    """The GPU stand-in for VanillaBootstrap (src/bootstrap/vanilla.rs:22-69):
    blind rotate -> sample_extract_index(.,0) -> identity_key_switching."""

    def __init__(self, device: int = 0):
        self.device = device

    def _engine(self, cloud_key) -> Engine:
        eng = engine_for(_params_of(cloud_key), self.device)
        eng.ensure_key(cloud_key)
        return eng

    def bootstrap(self, ctxt, cloud_key):  # vanilla.rs:40-52
        ctxt = np.asarray(ctxt, dtype=np.uint32)
        out = self._engine(cloud_key).batch_bootstrap(ctxt, None, True)
        return out[0] if ctxt.ndim == 1 else out

    def bootstrap_without_key_switch(self, ctxt, cloud_key):  # vanilla.rs:54-63
        ctxt = np.asarray(ctxt, dtype=np.uint32)
        out = self._engine(cloud_key).batch_bootstrap(ctxt, None, False)
        return out[0] if ctxt.ndim == 1 else out

    def name(self) -> str:  # vanilla.rs:65-67 returns "vanilla"
        return "hip-gfx950"


class LutBootstrap(HipBootstrap):
    """src/bootstrap/lut.rs:24-126"""

    def bootstrap_func(self, ct_in, f, message_modulus: int, cloud_key):  # lut.rs:49-65
        lut = Generator(message_modulus).generate_lookup_table(f)
        return self.bootstrap_lut(ct_in, lut, cloud_key)

    def bootstrap_lut(self, ct_in, lut: LookupTable, cloud_key):  # lut.rs:79-99
        ct_in = np.asarray(ct_in, dtype=np.uint32)
        out = self._engine(cloud_key).batch_bootstrap(ct_in, lut.poly, True)
        return out[0] if ct_in.ndim == 1 else out

    def bootstrap(self, ctxt, cloud_key):  # lut.rs:108-111: identity function, m = 2
        return self.bootstrap_func(ctxt, lambda x: x, 2, cloud_key)

    def bootstrap_without_key_switch(self, ctxt, cloud_key):  # lut.rs:113-121: just bootstrap
        return self.bootstrap(ctxt, cloud_key)

    def name(self) -> str:
        return "lut-hip-gfx950"


def default_bootstrap() -> Bootstrap:  # src/bootstrap/mod.rs:41-43
    return HipBootstrap()
