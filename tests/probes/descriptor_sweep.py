#!/usr/bin/env python3
"""Integer histograms of both descriptor kernels against the oracle for EVERY cloud size 2..260 (tile of 16 queries, neighbour
segments, the lone last tile), random row blocks, 2 and 4 frames, generic and lattice clouds; and the chi-square matrices built
from them (all eight, bit patterns) for every pair of sizes drawn.  Usage: python tests/probes/descriptor_sweep.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import oracle  # noqa: E402
from platymatch_amd import _kernels as K, _native as nat  # noqa: E402

oracle.build(); nat.load()
dev = torch.device("cuda:0")


def d(x):
    return nat.to_dev(np.ascontiguousarray(x, dtype=np.float64), dev=dev)


bad = []
rng = np.random.default_rng(11)
prev = None
checked = costs = 0
for n in list(range(2, 261)) + [511, 512, 513, 1023, 1025, 2049, 4099]:
    x = rng.normal(size=(3, n)) * rng.uniform(5, 60, size=(3, 1)) + rng.uniform(-200, 200, size=(3, 1))
    if n % 5 == 0:
        x = np.round(x * 0.5) * 2.0
    c, md, x0 = oracle.get_centroid(x, False), oracle.get_mean_distance(x, False), oracle.pca_axis(x.T)
    if not np.isfinite(md) or md == 0.0:
        continue
    nf = 4 if n % 2 else 2
    oc, ot = oracle.shape_context_counts(c, md, x, "fixed" if nf == 4 else "moving", x0=x0)
    row0 = int(rng.integers(0, n))
    nrows = int(rng.integers(1, n - row0 + 1))
    for path in ("tiled", "general"):
        for (r0, nr) in ((0, None), (row0, nrows)):
            r = K.shape_context(d(x), d(np.ravel(c)), d(x0), d(np.array([md])), nf, row0=r0, nrows=nr, want_counts=True, want_hist=True, path=path)
            sl = slice(r0, None if nr is None else r0 + nr)
            if not (np.array_equal(r["counts"].cpu().numpy(), oc[:, sl]) and np.array_equal(r["totals"].cpu().numpy(), ot[:, sl])):
                bad.append("histograms N=%d %s rows [%d, %s)" % (n, path, r0, "end" if nr is None else r0 + nr))
            elif not np.array_equal(r["hist"].cpu().numpy(), oracle.normalise_counts(oc[:, sl], ot[:, sl]), equal_nan=True):
                bad.append("normalised descriptors N=%d %s" % (n, path))
    checked += 1
    um = oracle.normalise_counts(oc, ot)
    if prev is not None and um.shape[0] != prev.shape[0]:
        a, b = (um, prev) if um.shape[0] == 2 else (prev, um)
        U8 = K.chi2_cost8(d(a), d(b)).cpu().numpy()
        for h, nm in enumerate(oracle.HYPOTHESES):
            want = np.asarray(oracle.unary_distance_matrix(a[int(nm[0]) - 1], b[int(nm[1]) - 1]))
            if not np.array_equal(np.nan_to_num(U8[h], nan=-1.0).view(np.uint64), np.nan_to_num(want, nan=-1.0).view(np.uint64)):
                bad.append("cost matrix %s for %d x %d" % (nm, a.shape[1], b.shape[1]))
        costs += 1
    prev = um
print("clouds checked: %d, cost builds: %d; mismatches: %d" % (checked, costs, len(bad)))
for b in bad[:30]:
    print("  " + b)
sys.exit(1 if bad else 0)
