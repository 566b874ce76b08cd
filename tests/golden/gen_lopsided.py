#!/usr/bin/env python3
"""The unmodified reference's verdict on the lopsided lattice pairs of round 3's soak (profiles/r03_soak_parity_lopsided.txt):
one cloud of 4..12 points against one of up to 600, both on a half-integer lattice with duplicates — the eight seeds on which
the HIP path and the CPU oracle gave different assignments / inlier counts (every one of them with a non-zero edge guard), and
eight controls of the same family on which they agreed.  gen_golden.run_pipeline drives the reference's own functions in the
widget's stage order with the soak's parameters (80 trials, the soak's error rule, 4 ICP iterations, the case's RANSAC seed)
and the integer histograms, the eight assignment vectors, inlier counts, A_sc, every ICP correspondence and A_final are kept.
Data only; build container only.
Usage: python tests/golden/gen_lopsided.py"""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import gen_golden as G  # noqa: E402
from soak_cases import LOPSIDED_CONTROLS, LOPSIDED_DIFFERING, make_case_lopsided  # noqa: E402

ref = G.import_reference()
out = {}
seeds = list(LOPSIDED_DIFFERING) + list(LOPSIDED_CONTROLS)
for k, seed in enumerate(seeds):
    mv, fx, lattice, transform, rs = make_case_lopsided(seed, 600)
    assert lattice and transform == "Affine"
    err_rule = 25.0 * (np.abs(mv).max() / 300.0 + 1e-9)
    p = "s%d_" % seed
    out[p + "moving"], out[p + "fixed"] = mv, fx
    out[p + "ransac_seed"], out[p + "ransac_error"] = np.array([rs], dtype=np.int64), np.array([err_rule])
    with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
        try:
            r = G.run_pipeline(ref, mv, fx, 80, err_rule, 4, 1, "lopsided%d" % seed, seed=rs)
        except Exception as e:                      # (e.g. a NaN cost matrix: scipy refuses; recorded as such)
            r, err = None, repr(e)
    if r is None:
        out[p + "raised"] = np.array([err])
        print("seed %d (%d x %d): reference raised %s" % (seed, mv.shape[1], fx.shape[1], err), flush=True)
        continue
    for key in ("lsa_rows", "lsa_cols", "ransac_inliers", "ransac_A", "A_sc", "icp_nn", "A_icp", "A_final",
                "counts_m1", "counts_m2", "counts_f1", "counts_f2", "counts_f3", "counts_f4",
                "total_m1", "total_m2", "total_f1", "total_f2", "total_f3", "total_f4", "mean_dist_m", "mean_dist_f", "x0_m", "x0_f"):
        out[p + key] = r[key]
    print("seed %d (%d x %d): inliers %s" % (seed, mv.shape[1], fx.shape[1], list(r["ransac_inliers"])), flush=True)
out["seeds"] = np.array(seeds, dtype=np.int64)
out["differing_in_round3"] = np.array(LOPSIDED_DIFFERING, dtype=np.int64)
np.savez_compressed(os.path.join(HERE, "lopsided.npz"), **out)
print("wrote", os.path.getsize(os.path.join(HERE, "lopsided.npz")), "bytes")
