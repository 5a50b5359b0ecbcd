import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import pyoracle
    pyoracle.lib()      # builds on demand (g++ only)
    return pyoracle


@pytest.fixture(scope="session")
def gpu():
    """The C-ABI library, initialised on device 0. Fails loudly when the HIP build or the device is missing."""
    from mitsuba_amd import capi
    capi.init(0)
    return capi


@pytest.fixture(scope="session")
def small_scene():
    from mitsuba_amd import scene
    return scene.small_test_scene()


@pytest.fixture(scope="session")
def cornell_scene():
    from mitsuba_amd import scene
    return scene.cornell_box(96, 96)


def random_rays(sd, n, seed=0):
    """Random chords through the bounding sphere of the scene (the ray distribution of the reference's
    kd-tree benchmark, src/tests/test_kd.cpp:112-117), mint = Epsilon so the adaptive epsilon path is used."""
    rng = np.random.default_rng(seed)
    lo, hi = sd.positions.min(0), sd.positions.max(0)
    c, r = (lo + hi) / 2, np.linalg.norm(hi - lo) / 2
    a = rng.normal(size=(n, 3)); a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = rng.normal(size=(n, 3)); b /= np.linalg.norm(b, axis=1, keepdims=True)
    o = c + a * r
    d = (c + b * r) - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((n, 8), np.float32)
    rays[:, :3], rays[:, 3], rays[:, 4:7], rays[:, 7] = o, 1e-4, d, np.inf
    return rays


def surface_rays(sd, oracle_scene, n, seed=1):
    """Secondary rays starting on surfaces (exercise the epsilon offsets): hit points of random chords,
    re-emitted into random directions, half of them with a finite maxt like shadow rays."""
    rng = np.random.default_rng(seed)
    h = oracle_scene.trace(random_rays(sd, 3 * n, seed + 100))
    ok = h["prim"] != 0xFFFFFFFF
    p = h["p"][ok][:n]
    d = rng.normal(size=p.shape); d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((p.shape[0], 8), np.float32)
    rays[:, :3], rays[:, 3], rays[:, 4:7] = p, 1e-4, d
    rays[:, 7] = np.where(rng.uniform(size=p.shape[0]) < 0.5, np.inf, rng.uniform(0.5, 6.0, p.shape[0]))
    return rays
