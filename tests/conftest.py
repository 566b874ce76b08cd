import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
SCENARIOS = ["synth128", "synth96x128", "insitu02_identity", "insitu02_affine", "insitu04_affine", "synth1000"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_library_is_built():
    """The tests bind libplatymatch_hip.so (host solver, RNG replica, argument checks: no GPU needed for those): build it in-tree
    if a fresh checkout has not yet (hipcc cross-compiles without a GPU; __graft_entry__.build() does the same)."""
    from platymatch_amd.build import build_native
    build_native()


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def oracle():
    """The CPU checker (oracle/): built with gcc on first use.  Test infrastructure only."""
    import oracle as o
    o.build()
    return o


@pytest.fixture(scope="session")
def micro():
    return load_golden("micro")


@pytest.fixture(scope="session", params=SCENARIOS)
def scenario(request):
    return request.param, load_golden(request.param)


def synth_pair(n, seed, sigma=1.0, m=None):
    """BASELINE.md §3 synthetic recipe: anisotropic blob, fixed = A_gt . moving (+ jitter, permuted)."""
    A_gt = load_golden("synth128")["A_gt"]
    rng = np.random.default_rng(seed)
    mv = rng.normal(size=(3, n)) * np.array([[60.0], [40.0], [25.0]]) + 200.0
    hom = np.vstack([mv, np.ones((1, n))])
    fx = (A_gt @ hom)[:3] + rng.normal(scale=sigma, size=(3, n))
    fx = np.ascontiguousarray(fx[:, rng.permutation(n)])
    if m is not None:
        fx = np.ascontiguousarray(fx[:, :m])
    return np.ascontiguousarray(mv), fx, A_gt
