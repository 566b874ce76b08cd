#!/usr/bin/env python3
"""The oracle's get_shape_context with RANDOM binning arguments (oracle/pm_oracle.py: the literal NumPy restatement of
shape_context.py:10-58) against the UNMODIFIED reference's, bit for bit, on random Gaussian / lattice / special-vector neighbour
lists — what pins the generalised oracle beyond tests/golden/binning.npz.  Build container only (the reference is mounted at
/root/reference; import recipe of gen_golden.py); nothing here is used at test time.
Usage: python tests/golden/soak_binning_oracle_vs_reference.py [seconds] [seed]"""
import os
import sys
import time
import warnings

import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import gen_golden as G
import oracle
ref = G.import_reference()[0]
oracle.build()
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
rng = np.random.default_rng(seed)
warnings.simplefilter("ignore")
t_end = time.time() + seconds
cases = bad = 0
while time.time() < t_end:
    kind = int(rng.integers(3))
    n = int(rng.integers(1, 120))
    if kind == 0:
        nb = rng.normal(size=(n, 3)) * rng.uniform(0.01, 200.0, size=3)
    elif kind == 1:
        nb = rng.integers(-5, 6, size=(n, 3)).astype(np.float64) * float(rng.choice([1.0, 0.5, 3.0]))
    else:
        nb = rng.normal(size=(n, 3)); nb[rng.integers(0, n)] = [0, 0, -1]; nb[rng.integers(0, n)] = [-1, -0.0, 0]
    md = float(np.exp(rng.uniform(np.log(0.05), np.log(50.0))))
    r_in = float(np.exp(rng.uniform(np.log(0.02), np.log(2.0)))); r_out = r_in * float(np.exp(rng.uniform(0.0, np.log(40.0))))
    n_r, n_t, n_p = int(rng.integers(1, 9)), int(rng.integers(1, 17)), int(rng.integers(1, 33))
    with np.errstate(all="ignore"):
        a = np.asarray(ref.get_shape_context(nb, md, r_inner=r_in, r_outer=r_out, n_rbins=n_r, n_thetabins=n_t, n_phibins=n_p), dtype=np.float64)
        b = oracle.get_shape_context(nb, md, r_in, r_out, n_r, n_t, n_p)
    cases += 1
    if not (a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))):
        bad += 1
print("oracle.get_shape_context with random binning arguments against the unmodified reference (build container, %.0f s, seed %d): %d cases, %d mismatches" % (seconds, seed, cases, bad))
sys.exit(1 if bad else 0)
