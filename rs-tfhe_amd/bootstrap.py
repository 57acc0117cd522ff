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
import contextlib
import threading

import numpy as np

from .engine import COPY, Engine
from .lut import Generator, LookupTable
from .params import DEFAULT_SECURITY, SecurityParams

_engines: dict = {}  # (params, device) -> [Engine, ...]
_engines_mu = threading.Lock()
MAX_ENGINES_PER_SET = 4  # resident cloud keys per (parameter set, device): 172 MB each at SECURITY_128_BIT


def engine_for(params: SecurityParams, device: int = 0) -> Engine:
    """The first Engine (C-ABI context) of (parameter set, device) in this process."""
    with _engines_mu:
        pool = _engines.setdefault((params, device), [])
        if not pool:
            pool.append(Engine(params, device))
        return pool[0]


@contextlib.contextmanager
def keyed_engine(cloud_key, device: int = 0):
    """An Engine holding exactly `cloud_key`, locked for the duration of the `with` body.

    The reference passes `&CloudKey` into every call (bootstrap/mod.rs:23-38 is `Send + Sync`); a context holds
    one key at a time, so choosing the key and launching under it must be ONE critical section -- two threads
    with two keys would otherwise compute under each other's key.  Up to MAX_ENGINES_PER_SET keys stay resident
    per (parameter set, device), so alternating between a few keys does not re-upload them; beyond that the
    least recently used context takes the new key."""
    params = _params_of(cloud_key)
    with _engines_mu:
        pool = _engines.setdefault((params, device), [])
        eng = next((e for e in pool if e._key is cloud_key), None)
        if eng is None:
            eng = next((e for e in pool if e._key is None), None)
        if eng is None and len(pool) < MAX_ENGINES_PER_SET:
            eng = Engine(params, device)
            pool.append(eng)
        if eng is None:
            eng = min(pool, key=lambda e: e._last_use)
        eng._last_use = next(_ticks)
    with eng.lock:
        eng.ensure_key(cloud_key)
        yield eng


def _tick_counter():
    i = 0
    while True:
        i += 1
        yield i


_ticks = _tick_counter()


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

    def bootstrap(self, ctxt, cloud_key):  # vanilla.rs:40-52
        ctxt = np.asarray(ctxt, dtype=np.uint32)
        with keyed_engine(cloud_key, self.device) as eng:
            out = eng.batch_bootstrap(ctxt, None, True)
        return out[0] if ctxt.ndim == 1 else out

    def bootstrap_without_key_switch(self, ctxt, cloud_key):  # vanilla.rs:54-63
        ctxt = np.asarray(ctxt, dtype=np.uint32)
        with keyed_engine(cloud_key, self.device) as eng:
            out = eng.batch_bootstrap(ctxt, None, False)
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
        with keyed_engine(cloud_key, self.device) as eng:
            out = eng.batch_bootstrap(ct_in, lut.poly, True)
        return out[0] if ct_in.ndim == 1 else out

    def bootstrap(self, ctxt, cloud_key):  # lut.rs:108-111: identity function, m = 2
        return self.bootstrap_func(ctxt, lambda x: x, 2, cloud_key)

    def bootstrap_without_key_switch(self, ctxt, cloud_key):  # lut.rs:113-121: just bootstrap
        return self.bootstrap(ctxt, cloud_key)

    def name(self) -> str:
        return "lut-hip-gfx950"


def default_bootstrap() -> Bootstrap:  # src/bootstrap/mod.rs:41-43
    return HipBootstrap()
