#!/usr/bin/env python3
"""The soak's second family (tests/soak_cases.make_case_b): integer voxel coordinates, a cloud against itself, tiny clouds, planar
clouds, RANSAC samples of 3 / 5 / 8 pairs, supervised mode — complete registrations, HIP path against the CPU oracle; a refusal
(exception) must be the same on both sides.  Usage: python tests/probes/soak_parity_b.py [seconds] [max_points] [first_seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401
import oracle  # noqa: E402
import platymatch_amd  # noqa: E402
from platymatch_amd import _native as nat  # noqa: E402
from platymatch_amd.estimate_transform import perform_icp as pi  # noqa: E402
from soak_cases import make_case_b  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
max_points = int(sys.argv[2]) if len(sys.argv) > 2 else 400
seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 0
oracle.build(); nat.load(); pi.VERBOSE = False
KINDS = ["voxel", "self", "tiny", "planar", "k-samples", "supervised"]


def relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def same(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return a.shape == b.shape and np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.isinf(a), np.isinf(b))


counts = {k: 0 for k in KINDS}
counts.update(cases=0, both_raised=0, compared_transforms=0)
fails = []
t_end = time.perf_counter() + budget
t_print = time.perf_counter() + 30.0
seed = seed0
while time.perf_counter() < t_end:
    c = make_case_b(seed, max_points)
    mv, fx, kw = c["mv"], c["fx"], c["kwargs"]
    tag = "seed %d (%s, N=%d, M=%d, %s%s)" % (seed, KINDS[c["kind"]], mv.shape[1], fx.shape[1], kw["transform"],
                                                ", k=%d" % kw["ransac_samples"] if "ransac_samples" in kw else "")
    det, odet = {}, {}
    ref = got = ref_exc = got_exc = None
    try:
        ref = oracle.estimate_transform(mv, fx, details=odet, **kw)
    except Exception as e:
        ref_exc = e
    try:
        got = platymatch_amd.register(mv, fx, details=det, **kw)
    except Exception as e:
        got_exc = e
    if ref_exc is not None or got_exc is not None:
        if ref_exc is None or got_exc is None:
            fails.append(tag + ": oracle %s, product %s" % ("raised %s: %s" % (type(ref_exc).__name__, str(ref_exc)[:120]) if ref_exc else "ran",
                                                           "raised %s: %s" % (type(got_exc).__name__, str(got_exc)[:120]) if got_exc else "ran"))
        else:
            counts["both_raised"] += 1
    else:
        if "lsa" in odet:
            for h in range(8):
                if not (np.array_equal(det["lsa"][h][0], odet["lsa"][h][0]) and np.array_equal(det["lsa"][h][1], odet["lsa"][h][1])):
                    fails.append(tag + ": assignment of hypothesis %d differs" % h)
                    break
            if not np.array_equal(got[2], ref[2]):
                fails.append(tag + ": inlier counts differ %s vs %s" % (list(got[2]), list(ref[2])))
        finite = np.isfinite(ref[0]).all() and np.isfinite(ref[1]).all()
        if not (same(got[0], ref[0]) and same(got[1], ref[1])):
            fails.append(tag + ": non-finite patterns differ")
        elif finite and np.linalg.cond(ref[0]) < 1e8:
            if "nn" in odet and not np.array_equal(det["nn"], odet["nn"]):
                fails.append(tag + ": ICP correspondences differ (%d of %d)" % (int((np.asarray(det["nn"]) != np.asarray(odet["nn"])).sum()), np.asarray(odet["nn"]).size))
            e1, e2 = relerr(got[0], ref[0]), relerr(got[1] @ got[0], ref[1] @ ref[0])
            if not (e1 < 1e-6 and e2 < 1e-6):
                fails.append(tag + ": transforms differ (%.1e, %.1e)" % (e1, e2))
            counts["compared_transforms"] += 1
    counts[KINDS[c["kind"]]] += 1
    counts["cases"] += 1
    seed += 1
    if time.perf_counter() > t_print:
        print("... %d cases, %d mismatches" % (counts["cases"], len(fails)), flush=True)
        t_print = time.perf_counter() + 30.0
print("soak B: seeds %d..%d, up to %d points: %s" % (seed0, seed - 1, max_points, counts))
print("mismatches: %d" % len(fails))
for f in fails[:60]:
    print("  " + f)
sys.exit(1 if fails else 0)
