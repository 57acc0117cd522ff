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

_engines: dict = {}  # (params, device) -> Engine: ONE C-ABI context per parameter set and device
_views: dict = {}  # (params, device) -> {id(cloud_key): key view}: the resident keys of that context
_engines_mu = threading.Lock()
MAX_RESIDENT_KEYS = 4  # key views kept per (parameter set, device): 172 MB each at SECURITY_128_BIT (+ 104 MB of byte planes)


def engine_for(params: SecurityParams, device: int = 0) -> Engine:
    """THE Engine (C-ABI context) of (parameter set, device) in this process."""
    with _engines_mu:
        eng = _engines.get((params, device))
        if eng is None:
            eng = _engines[(params, device)] = Engine(params, device)
        return eng


def _evict_idle(views: dict, room_for: int = 1) -> None:
    """(_engines_mu held) Drop least recently used IDLE views until `room_for` more fit under MAX_RESIDENT_KEYS:
    each resident key pins 172 MB + 104 MB of byte planes on the device at SECURITY_128_BIT and, through the view's
    reference to its CloudKey, ~160 MB of host memory."""
    idle = [k for k, v in views.items() if v._users == 0]
    while len(views) + room_for > MAX_RESIDENT_KEYS and idle:
        victim = min(idle, key=lambda k: views[k]._last_use)
        idle.remove(victim)
        views.pop(victim).close()


def adopt_view(cloud_key, view: Engine) -> None:
    """Register a key view that already holds `cloud_key` (e.g. the one it was generated in): first use uploads nothing.
    Counts against MAX_RESIDENT_KEYS like any other view: a loop of SecretKey.cloud_key() calls (key rotation, test
    suites) keeps at most that many keys resident instead of growing without bound."""
    with _engines_mu:
        views = _views.setdefault((view.params, view.device), {})
        stale = views.pop(id(cloud_key), None)
        if stale is not None and stale is not view and stale._users == 0:
            stale.close()
        _evict_idle(views)
        view._key = cloud_key
        view._last_use = next(_ticks)
        views[id(cloud_key)] = view


@contextlib.contextmanager
def keyed_engine(cloud_key, device: int = 0):
    """The key view of `cloud_key` on the one context of its (parameter set, device).

    The reference passes `&CloudKey` into every call (bootstrap/mod.rs:23-38, `Send + Sync`).  Here a key is a KEY
    VIEW of the context (`tfhe_hip_key_create`, include/tfhe_hip.h): its own resident key, the context's device,
    streams, scratch and mutex.  A call names its key by the handle it is made on, so two threads with two keys
    share one context and cannot compute under each other's key; up to MAX_RESIDENT_KEYS views stay resident per
    (parameter set, device), beyond that the least recently used idle view is dropped."""
    params = _params_of(cloud_key)
    base = engine_for(params, device)
    with _engines_mu:
        views = _views.setdefault((params, device), {})
        view = views.get(id(cloud_key))
        if view is not None and view._key is not cloud_key:  # a recycled id(): not this key
            view = None
        fresh = view is None
        if fresh:
            _evict_idle(views)
            view = base.new_key_view()
            view._key = cloud_key  # held: the id cannot be recycled while the view lives
            views[id(cloud_key)] = view
        view._users += 1
        view._last_use = next(_ticks)
    try:
        if fresh or not view._lib.tfhe_hip_key_is_loaded(view._ctx):
            with view.lock:  # two threads meeting on a fresh view: one uploads
                if not view._lib.tfhe_hip_key_is_loaded(view._ctx):
                    view.load_cloud_key(cloud_key)
        yield view
    finally:
        with _engines_mu:
            view._users -= 1


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
