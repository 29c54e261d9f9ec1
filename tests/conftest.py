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


def random_graph(po, f, rng, n_calcs, n_fixed, n_advice, n_instance, n_chal, long_lived):
    """A random straight-line program over every source kind and operation.  long_lived > 0 keeps
    that many early intermediates alive to the end, forcing slots beyond the LDS budget into HBM."""
    consts = [0, 1, 2] + [rng.below(f.p) for _ in range(4)]
    rotations = [0, 1, -1, 2, -3, 5]
    calcs, n_int = [], 0

    def src():
        kinds = [po.SRC_CONSTANT, po.SRC_FIXED, po.SRC_ADVICE, po.SRC_ADVICE, po.SRC_CHALLENGE, po.SRC_BETA, po.SRC_GAMMA, po.SRC_THETA, po.SRC_Y, po.SRC_PREVIOUS]
        if n_instance: kinds.append(po.SRC_INSTANCE)
        if n_int: kinds += [po.SRC_INTERMEDIATE] * 8
        k = kinds[rng.below(len(kinds))]
        if k == po.SRC_CONSTANT: return (k, rng.below(len(consts)), 0)
        if k == po.SRC_INTERMEDIATE: return (k, rng.below(n_int), 0)
        if k == po.SRC_FIXED: return (k, rng.below(n_fixed), rng.below(len(rotations)))
        if k == po.SRC_ADVICE: return (k, rng.below(n_advice), rng.below(len(rotations)))
        if k == po.SRC_INSTANCE: return (k, rng.below(n_instance), rng.below(len(rotations)))
        if k == po.SRC_CHALLENGE: return (k, rng.below(n_chal), 0)
        return (k, 0, 0)

    for _ in range(n_calcs):
        op = rng.below(8)
        parts = tuple(src() for _ in range(1 + rng.below(4))) if op == po.CALC_HORNER else ()
        calcs.append((op, src(), src(), parts, n_int))
        n_int += 1
    if long_lived:   # a final Horner over the first `long_lived` intermediates keeps all of them live
        calcs.append((po.CALC_HORNER, (po.SRC_INTERMEDIATE, n_int - 1, 0), (po.SRC_Y, 0, 0), tuple((po.SRC_INTERMEDIATE, i, 0) for i in range(long_lived)), n_int))
        n_int += 1
    return {"constants": consts, "rotations": rotations, "calcs": calcs, "num_intermediates": n_int}


