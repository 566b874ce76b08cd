#!/usr/bin/env python3
"""The unmodified reference END TO END (gen_golden.run_pipeline: statistics -> descriptors -> eight cost matrices -> eight
assignments -> seeded RANSAC -> ICP) on 24 small pairs: generic, integer voxel coordinates, a coarse lattice with duplicates,
and integer x, y with a float z (the layout of the reference's assets).  Keeps what a registration is judged on: the eight
assignment vectors, inlier counts, A_sc, every ICP correspondence, A_final.  Data only; build container only.
Usage: python tests/golden/gen_random_e2e.py"""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402

ref = G.import_reference()
out = {}
CASES = 24
for k in range(CASES):
    rng = np.random.default_rng(929292 * k + 5)
    n, m = int(rng.integers(14, 36)), int(rng.integers(14, 36))
    if k % 3 == 0:
        m = n
    base = rng.normal(size=(3, max(n, m))) * rng.uniform(10, 40, size=(3, 1)) + rng.uniform(20, 250, size=(3, 1))
    A = np.eye(3) + rng.normal(scale=0.08, size=(3, 3))
    mv = base.copy()
    fx = A @ base + rng.uniform(-15, 15, size=(3, 1)) + rng.normal(scale=0.4, size=base.shape)
    kind = k % 4
    if kind == 1:
        mv, fx = np.round(mv), np.round(fx)
    elif kind == 2:
        mv, fx = np.round(mv * 0.2) * 5.0, np.round(fx * 0.2) * 5.0
    elif kind == 3:
        mv[:2], fx[:2] = np.round(mv[:2]), np.round(fx[:2])
    mv = np.ascontiguousarray(mv[:, :n])
    fx = np.ascontiguousarray(fx[:, rng.permutation(max(n, m))[:m]])
    with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
        try:
            r = G.run_pipeline(ref, mv, fx, 60, 12.0, 4, 1, "e2e%d" % k)
        except Exception as e:                      # (a NaN cost matrix: scipy refuses; recorded as such)
            r = None
            err = repr(e)
    p = "c%02d_" % k
    out[p + "moving"], out[p + "fixed"], out[p + "kind"] = mv, fx, np.array([kind])
    if r is None:
        out[p + "raised"] = np.array([err])
        print("case %d (%d x %d, kind %d): reference raised %s" % (k, n, m, kind, err), flush=True)
        continue
    for key in ("lsa_rows", "lsa_cols", "ransac_inliers", "A_sc", "icp_nn", "A_icp", "A_final", "ransac_A"):
        out[p + key] = r[key]
    print("case %d (%d x %d, kind %d): inliers %s" % (k, n, m, kind, list(r["ransac_inliers"])), flush=True)
out["cases"] = np.array([CASES])
np.savez_compressed(os.path.join(HERE, "random_e2e.npz"), **out)
print("wrote", os.path.getsize(os.path.join(HERE, "random_e2e.npz")), "bytes")
