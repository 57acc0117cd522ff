"""rs-tfhe_amd: MI355X-native (gfx950 HIP) engine for the gate-bootstrapping hot
path of thedonutfactory/rs-tfhe, behind the reference's own API surface.

    from rs_tfhe_amd import gates, key, params, bootstrap, lut, client

The compute is entirely in rs-tfhe_amd/libtfhe_hip.so (C ABI in
include/tfhe_hip.h); importing this package never falls back to a CPU path.
"""
from . import _capi, bootstrap, circuit, client, distributed, engine, gates, key, lut, params, proxy_reenc  # noqa: F401
from .bootstrap import Bootstrap, HipBootstrap, LutBootstrap, default_bootstrap  # noqa: F401
from .circuit import Circuit  # noqa: F401
from .client import SecretKey  # noqa: F401
from .engine import Engine, Pool  # noqa: F401
from .gates import Gates  # noqa: F401
from .key import CloudKey  # noqa: F401
from .params import SECURITY_128_BIT, SecurityParams  # noqa: F401

__all__ = ["Engine", "Pool", "Gates", "CloudKey", "Bootstrap", "HipBootstrap", "LutBootstrap", "default_bootstrap",
           "SecurityParams", "SECURITY_128_BIT", "gates", "key", "params", "bootstrap", "lut", "engine",
           "distributed", "circuit", "Circuit", "client", "SecretKey", "proxy_reenc"]
