#!/usr/bin/env python3
"""A lopsided soak case in which the product raised and the oracle did not: where, and were the stages before it equal?
Usage: PM_SOAK_LOPSIDED=1 python tests/probes/soak_debug_raise.py SEED"""
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
import platymatch_amd  # noqa: E402
from platymatch_amd import _native as nat  # noqa: E402
from platymatch_amd.estimate_transform import perform_icp as pi  # noqa: E402
from soak_cases import make_case  # noqa: E402

seed = int(sys.argv[1])
oracle.build(); nat.load(); pi.VERBOSE = False
mv, fx, lattice, transform, rs = make_case(seed, 600)
k = 4 + seed % 9
if seed % 2:
    fx = np.ascontiguousarray(fx[:, :k])
else:
    mv = np.ascontiguousarray(mv[:, :k])
err = 25.0 * (np.abs(mv).max() / 300.0 + 1e-9)
print("N=%d M=%d lattice=%s %s; unique moving points %d" % (mv.shape[1], fx.shape[1], lattice, transform, len(np.unique(mv.T, axis=0))))
odet, det = {}, {}
ref = oracle.estimate_transform(mv, fx, transform=transform, ransac_trials=80, ransac_error=err, icp_iterations=4, seed=rs, details=odet)
print("oracle: inliers", list(ref[2]), "A_sc finite", np.isfinite(ref[0]).all(), "A_icp finite", np.isfinite(ref[1]).all())
try:
    got = platymatch_amd.register(mv, fx, transform=transform, ransac_trials=80, ransac_error=err, icp_iterations=4, seed=rs, details=det)
    print("product ran: inliers", list(got[2]))
except Exception:
    traceback.print_exc()
print("edge guard", det.get("edge_guard"))
if "lsa" in det:
    print("assignments equal:", [bool(np.array_equal(det["lsa"][h][1], odet["lsa"][h][1])) for h in range(8)])
if "ransac_A" in det:
    print("ransac_A finite per hypothesis:", [bool(np.isfinite(a).all()) for a in det["ransac_A"]])
