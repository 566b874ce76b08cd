#!/usr/bin/env python3
"""Reference outputs for a set of small RANDOM clouds (beyond the six scenarios of gen_golden.py): what the unmodified reference
computes for 36 generic pairs of 8-40 points — centroid, mean distance, all six descriptor sets (as integer counts), two of the
eight chi-square matrices in full, seeded do_ransac (both modes) and a 5-iteration perform_icp (both modes).  Data only; runs
only in the build container (reference at /root/reference).  Usage: python tests/golden/gen_random_small.py"""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import import_reference  # noqa: E402

sc, ft, at, icp, ut = import_reference()
out = {}
CASES = 36
for k in range(CASES):
    rng = np.random.default_rng(515151 * k + 9)
    n, m = int(rng.integers(8, 41)), int(rng.integers(8, 41))
    base = rng.normal(size=(3, max(n, m))) * rng.uniform(5, 40, size=(3, 1)) + rng.uniform(0, 200, size=(3, 1))
    A = np.eye(3) + rng.normal(scale=0.1, size=(3, 3))
    mv = np.ascontiguousarray(base[:, :n])
    fx = np.ascontiguousarray((A @ base + rng.uniform(-20, 20, size=(3, 1)) + rng.normal(scale=0.5, size=base.shape))[:, rng.permutation(max(n, m))[:m]])
    p = "c%02d_" % k
    out[p + "moving"], out[p + "fixed"] = mv, fx
    cm, cf = ut.get_centroid(mv, transposed=False), ut.get_centroid(fx, transposed=False)
    mdm, mdf = ut.get_mean_distance(mv, transposed=False), ut.get_mean_distance(fx, transposed=False)
    out[p + "centroid_m"], out[p + "centroid_f"] = np.asarray(cm), np.asarray(cf)
    out[p + "mean_dist"] = np.array([mdm, mdf])
    um = sc.get_unary(cm, mean_distance=mdm, detections=mv, type="moving", transposed=False)[:2]
    uf = sc.get_unary(cf, mean_distance=mdf, detections=fx, type="fixed", transposed=False)[:4]
    out[p + "counts_m"] = np.stack([np.rint(np.asarray(u) * (n - 1)) for u in um]).astype(np.int16)
    out[p + "counts_f"] = np.stack([np.rint(np.asarray(u) * (m - 1)) for u in uf]).astype(np.int16)
    for name, a, b in (("U11", um[0], uf[0]), ("U24", um[1], uf[3])):
        U = np.zeros((n, m))
        for i in range(n):
            for j in range(m):
                U[i, j] = sc.get_unary_distance(np.asarray(a)[i], np.asarray(b)[j])
        out[p + name] = U
    pq = min(n, m)
    rs = int(rng.integers(0, 2 ** 31))
    out[p + "seed"] = np.array([rs])
    for tr in ("Affine", "Similar"):
        np.random.seed(rs)
        A_r, inl = sc.do_ransac(mv[:, :pq], fx[:, :pq], 4, 30, 10.0, tr)
        out[p + "ransac_A_" + tr], out[p + "ransac_inl_" + tr] = np.asarray(A_r), np.array([inl])
        with contextlib.redirect_stdout(io.StringIO()):
            out[p + "icp_" + tr] = np.asarray(icp.perform_icp(mv.copy(), fx.copy(), 5, tr))
    print("case %d: N=%d M=%d" % (k, n, m), flush=True)
out["cases"] = np.array([CASES])
np.savez_compressed(os.path.join(HERE, "random_small.npz"), **out)
print("wrote", os.path.join(HERE, "random_small.npz"), os.path.getsize(os.path.join(HERE, "random_small.npz")), "bytes")
