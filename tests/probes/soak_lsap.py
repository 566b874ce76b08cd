#!/usr/bin/env python3
"""solve_on_device against scipy.optimize.linear_sum_assignment on a stream of random matrices: eleven families (normal, offset,
tiny scale, heavy tails, low rank, rounded = ties, 1-D geometry, negative, chi-square matrices of random histograms with repeated
rows, entries of +inf, a NaN), sizes 1..1500, wide and tall; whatever route is taken the indices are SciPy's, and where SciPy
raises the product raises the same error.  Usage: python tests/probes/soak_lsap.py [seconds] [first_seed]"""
import os
import sys
import time

import numpy as np
from scipy.optimize import linear_sum_assignment as scipy_lsa

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from platymatch_amd import _kernels as K, _native as nat, lsap as L  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
nat.load()
dev = torch.device("cuda:0")
routes, fails = {}, []
t_end = time.perf_counter() + budget
seed = seed0
while time.perf_counter() < t_end:
    rng = np.random.default_rng(777 * seed + 13)
    n, m = int(rng.integers(1, 1501)), int(rng.integers(1, 1501))
    if seed % 3 == 0:
        m = n
    kind = seed % 11
    if kind == 0:
        U = rng.normal(size=(n, m))
    elif kind == 1:
        U = rng.random((n, m)) + 1e6
    elif kind == 2:
        U = rng.random((n, m)) * 1e-200
    elif kind == 3:
        U = rng.standard_cauchy(size=(n, m))
    elif kind == 4:
        U = np.outer(rng.random(n), rng.random(m)) + 1e-3 * rng.random((n, m))
    elif kind == 5:
        U = np.round(rng.random((n, m)), 2)
    elif kind == 6:
        U = np.abs(rng.normal(size=(n, 1)) - rng.normal(size=(1, m)))
    elif kind == 7:
        U = -rng.random((n, m)) ** 3
    elif kind == 8:                                   # chi-square of histogram-like rows, some rows repeated (exact ties, zeros)
        p = rng.dirichlet(np.full(360, 0.3))
        a = rng.multinomial(200, p, size=n).astype(np.float64); a /= a.sum(1, keepdims=True)
        b = rng.multinomial(200, p, size=m).astype(np.float64); b /= b.sum(1, keepdims=True)
        if n > 3:
            a[rng.integers(0, n, size=n // 10)] = a[0]
        U = K.chi2_cost(nat.to_dev(a, dev=dev), nat.to_dev(b, dev=dev)).cpu().numpy()
    elif kind == 9:
        U = rng.random((n, m))
        U[rng.random((n, m)) < 0.05] = np.inf
    else:
        U = rng.random((n, m))
        if n * m > 1:
            U[int(rng.integers(0, n)), int(rng.integers(0, m))] = np.nan
    tag = "seed %d kind %d (%d x %d)" % (seed, kind, n, m)
    want = got = werr = gerr = None
    try:
        want = scipy_lsa(U)
    except Exception as e:
        werr = e
    info = {}
    try:
        got = L.solve_on_device(nat.to_dev(U, dev=dev), info=info, force=True)
    except Exception as e:
        gerr = e
    if werr is not None or gerr is not None:
        if werr is None or gerr is None or type(werr) is not type(gerr) or str(werr) != str(gerr):
            fails.append(tag + ": scipy %r, product %r" % (werr, gerr))
        routes["raised"] = routes.get("raised", 0) + 1
    else:
        if not (np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])):
            fails.append(tag + ": indices differ (route %s)" % info.get("route"))
        routes[info.get("route")] = routes.get(info.get("route"), 0) + 1
    seed += 1
print("lsap soak: seeds %d..%d, routes %s; mismatches: %d" % (seed0, seed - 1, routes, len(fails)))
for f in fails[:30]:
    print("  " + f)
sys.exit(1 if fails else 0)
