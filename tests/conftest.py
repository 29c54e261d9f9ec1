import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as entry  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return entry.load_package()


@pytest.fixture(scope="session")
def oracles():
    return entry.load_oracle()


@pytest.fixture(scope="session")
def po(oracles):
    return oracles[0]


@pytest.fixture(scope="session")
def co(oracles):
    return oracles[1]


@pytest.fixture(scope="session")
def ctx(pkg):
    """One device context for the whole GPU session.  No fallback: fails if the HIP library is
    not built or no gfx950 device is visible."""
    c = pkg.Context(0)
    yield c
    c.close()


def golden(name):
    with open(os.path.join(ROOT, "tests", "golden", name + ".json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def golden_loader():
    return golden


def enc_points(field_spec, pts):
    """[(x, y) | None] canonical ints -> n x 8 u64 Montgomery (identity = zeros)."""
    out = np.zeros((len(pts), 8), dtype=np.uint64)
    for i, P in enumerate(pts):
        if P is None:
            continue
        out[i, :4] = field_spec.encode(P[0])
        out[i, 4:] = field_spec.encode(P[1])
    return out


def dec_point(field_spec, xy):
    xy = np.asarray(xy).reshape(8)
    if not xy.any():
        return None
    return (field_spec.decode(xy[:4]), field_spec.decode(xy[4:]))
