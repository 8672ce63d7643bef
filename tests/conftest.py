import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


_FOAMS = {}


def small_foam(n_points, sh_degree, seed):
    """Cached synthetic foam (numpy dict); Qhull on a few thousand points takes well under a second."""
    from radfoam_amd import foam

    key = (n_points, seed)
    if key not in _FOAMS:
        _FOAMS[key] = foam.make_synthetic_foam(n_points, 0, seed)
    base = _FOAMS[key]
    rng = np.random.default_rng(1000 + seed * 7 + sh_degree)
    a = foam.attribute_dim(sh_degree)
    attrs = np.empty((n_points, a), dtype=np.float32)
    attrs[:, : a - 1] = rng.normal(0.0, 0.3, size=(n_points, a - 1)).astype(np.float32)
    attrs[:, a - 1] = base["attributes"][:, -1]
    out = dict(base)
    out["attributes"] = attrs
    out["sh_degree"] = sh_degree
    return out


@pytest.fixture(scope="session")
def foam_factory():
    return small_foam
