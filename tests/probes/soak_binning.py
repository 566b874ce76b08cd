#!/usr/bin/env python3
"""get_shape_context with random NON-DEFAULT binning arguments on the HIP path against the oracle's literal NumPy restatement of
shape_context.py:10-58 (same host, same NumPy: what is exercised is the device arithmetic, the host step tables and the hand-over of
neighbours on azimuth steps), bit for bit, for SECONDS seconds: Gaussian clouds of random scale, integer and half-integer lattices
(neighbours on sector planes, axes, diagonals), clouds with a few NaN / zero / axis vectors mixed in.
Usage: python tests/probes/soak_binning.py [seconds] [seed]"""
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from platymatch_amd import _native as nat  # noqa: E402
from platymatch_amd.estimate_transform import binning as B  # noqa: E402
from platymatch_amd.estimate_transform.shape_context import get_shape_context  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
oracle.build()
nat.load()
rng = np.random.default_rng(seed)
warnings.simplefilter("ignore")
t_end = time.time() + seconds
cases = bad = listed_cases = 0
kinds = {}
while time.time() < t_end:
    kind = ("gauss", "lattice", "half_lattice", "mixed")[int(rng.integers(4))]
    n = int(rng.integers(5, 400))
    if kind == "gauss":
        nb = rng.normal(size=(n, 3)) * rng.uniform(0.01, 200.0, size=3)
    elif kind in ("lattice", "half_lattice"):
        nb = rng.integers(-6, 7, size=(n, 3)).astype(np.float64) * (0.5 if kind == "half_lattice" else 1.0) * float(rng.choice([1.0, 3.0, 0.25]))
    else:
        nb = rng.normal(size=(n, 3)) * 10.0
        special = np.array([[0, 0, 0], [0, 0, 1], [0, 0, -1], [1, 0, 0], [-1, 0, 0], [-1, -0.0, 0], [0, 1, 0], [0, -1, 0], [np.nan, 1, 1],
                            [1, -1e-30, 0], [1, 1, 0], [-1, 1, 0], [1, 1, 1]], dtype=np.float64)
        nb[rng.integers(0, n, size=len(special))] = special
    md = float(np.exp(rng.uniform(np.log(0.05), np.log(50.0))))
    r_in = float(np.exp(rng.uniform(np.log(0.02), np.log(2.0))))
    r_out = r_in * float(np.exp(rng.uniform(0.0, np.log(40.0))))
    n_r, n_t, n_p = int(rng.integers(1, 9)), int(rng.integers(1, 17)), int(rng.integers(1, 33))
    if (r_in, r_out, n_r, n_t, n_p) == (1 / 8, 2, 5, 6, 12):
        continue
    with np.errstate(all="ignore"):
        want = oracle.get_shape_context(nb, md, r_in, r_out, n_r, n_t, n_p)
        got = get_shape_context(nb, md, r_in, r_out, n_r, n_t, n_p)
    cases += 1
    kinds[kind] = kinds.get(kind, 0) + 1
    if not (got.shape == want.shape and np.array_equal(got.view(np.uint64), want.view(np.uint64))):
        bad += 1
        if bad <= 10:
            print("MISMATCH kind %s n %d md %r binning %r: bins %s" % (kind, n, md, (r_in, r_out, n_r, n_t, n_p), np.flatnonzero(got != want)[:8]), flush=True)
print("seed %d, %.0f s: %d cases %s, %d mismatches; step tables cached for %d theta / %d phi bin counts"
      % (seed, seconds, cases, kinds, bad, B.cos_steps.cache_info().currsize, B.phi_steps.cache_info().currsize))
sys.exit(1 if bad else 0)
