import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


def _has_gpu() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def O():
    from oracle import oracle

    oracle.build()
    oracle.lib()
    return oracle


_KEYS = {}


def oracle_keys(O, params, seed=1234, with_time=False):
    k = (params.name, seed, with_time)
    if k not in _KEYS:
        _KEYS[k] = O.keygen(params, seed, with_time_domain=with_time)
    return _KEYS[k]


@pytest.fixture(scope="session")
def keys128(O):
    return oracle_keys(O, O.SECURITY_128_BIT, with_time=True)


@pytest.fixture(scope="session")
def keys80(O):
    return oracle_keys(O, O.SECURITY_80_BIT)


@pytest.fixture(scope="session")
def keys_uint4(O):
    return oracle_keys(O, O.SECURITY_UINT4, with_time=True)


@pytest.fixture(scope="session")
def golden():
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    return {
        "stage": np.load(os.path.join(d, "stage_vectors.npz")),
        "toy": np.load(os.path.join(d, "toy_bootstrap.npz")),
    }


def signed_diff(a, b):
    """max |a - b| on the torus (wrapping), as integers."""
    d = (np.asarray(a, np.uint32) - np.asarray(b, np.uint32)).astype(np.int32)
    return int(np.abs(d.astype(np.int64)).max()) if d.size else 0
