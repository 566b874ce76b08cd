#!/usr/bin/env python3
"""The reference's histograms for small clouds on integer voxel coordinates and on a coarse lattice (neighbours exactly on bin
boundaries, duplicates): the input the edge guard exists for.  Data only (clouds + integer counts, NaN rows marked); runs only in
the build container.  Usage: python tests/golden/gen_random_lattice.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import import_reference  # noqa: E402

sc, ft, at, icp, ut = import_reference()
out = {}
CASES = 90
for k in range(CASES):
    rng = np.random.default_rng(717171 * k + 11)
    n = int(rng.integers(8, 45))
    x = rng.normal(size=(3, n)) * rng.uniform(8, 40, size=(3, 1)) + rng.uniform(0, 250, size=(3, 1))
    kind = k % 3                                     # 0 voxel, 1 coarse lattice, 2 voxel x, y with a float z (like the reference's assets)
    if kind == 0:
        x = np.round(x)
    elif kind == 1:
        x = np.round(x * 0.2) * 5.0
    else:
        x[:2] = np.round(x[:2])
    x = np.ascontiguousarray(x)
    c, md = ut.get_centroid(x, transposed=False), ut.get_mean_distance(x, transposed=False)
    with np.errstate(all="ignore"):
        u = sc.get_unary(c, mean_distance=md, detections=x, type="fixed", transposed=False)[:4]
    u = np.stack([np.asarray(v) for v in u])                       # [4, n, 360], rows sum to 1 (or NaN)
    tot = np.zeros((4, n), dtype=np.int16)
    cnt = np.zeros((4, n, 360), dtype=np.int16)
    for f in range(4):
        for i in range(n):
            row = u[f, i]
            if np.isnan(row).any():
                tot[f, i] = -1
                continue
            pos = row[row > 0]
            t = int(round(1.0 / pos.min())) if pos.size else 0         # smallest positive value = 1 / total ... or k / total
            # totals are at most n - 1: find the smallest total that makes every entry an integer count
            for cand in range(1, n):
                if np.all(np.abs(row * cand - np.rint(row * cand)) < 1e-9):
                    t = cand
                    break
            tot[f, i] = t
            cnt[f, i] = np.rint(row * t).astype(np.int16)
    p = "c%02d_" % k
    out[p + "cloud"], out[p + "counts"], out[p + "totals"] = x, cnt, tot
out["cases"] = np.array([CASES])
np.savez_compressed(os.path.join(HERE, "random_lattice.npz"), **out)
print("wrote", os.path.getsize(os.path.join(HERE, "random_lattice.npz")), "bytes")
