#!/usr/bin/env python3
"""Many random registrations, HIP path against the CPU oracle, stage by stage — the fixed-seed parity tests of tests/test_gpu_*.py
run over a stream of fresh clouds for a time budget (a probe, not a test: it lives under tests/ because it calls the oracle).
Per case (random sizes N != M, anisotropy, offset, scale, jitter; every fifth cloud on a half-integer lattice, which puts
neighbours exactly on ring and sector edges and makes distances tie):
  statistics    centroid, mean distance and the product's PCA axis == oracle bits (the device's Jacobi axis kernel: within 1e-12, reported)
  descriptors   integer histograms (2 + 4 frames, tile path and general path) == oracle
  costs         the eight chi-square matrices == oracle float64 bit patterns
  registration  assignment vectors of all eight hypotheses, RANSAC inlier counts, every ICP correspondence == oracle;
                final 4x4 within 1e-6 relative (Affine and Similar in turn)
Usage: python tests/probes/soak_parity.py [seconds] [max_points] [first_seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import oracle  # noqa: E402
import platymatch_amd  # noqa: E402
from platymatch_amd import _kernels as K, _native as nat  # noqa: E402
from platymatch_amd.estimate_transform import perform_icp as pi, shape_context as SC  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
max_points = int(sys.argv[2]) if len(sys.argv) > 2 else 500
seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 0
oracle.build()
nat.load()
pi.VERBOSE = False
dev = torch.device("cuda:0")


def d(x):
    return nat.to_dev(np.ascontiguousarray(x, dtype=np.float64), dev=dev)


from soak_cases import make_case as _make_case, make_case_lopsided  # noqa: E402


def make_case(seed):
    if os.environ.get("PM_SOAK_LOPSIDED") == "1":        # one cloud cut down to 4..12 points: N >> M and N << M
        return make_case_lopsided(seed, max_points)
    return _make_case(seed, max_points)


def relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


counts = dict(cases=0, lattice=0, statistics=0, descriptors=0, costs=0, registrations=0, similar=0, skipped_registration=0, guard_hits=0)
fails = []
t_end = time.perf_counter() + budget
t_print = time.perf_counter() + 30.0
seed = seed0
while time.perf_counter() < t_end:
    mv, fx, lattice, transform, rs = make_case(seed)
    tag = "seed %d (N=%d, M=%d%s, %s)" % (seed, mv.shape[1], fx.shape[1], ", lattice" if lattice else "", transform)
    try:
        desc = {}
        for cloud, typ, nf in ((mv, "moving", 2), (fx, "fixed", 4)):
            x = d(cloud)
            c_o, md_o, x0_o = oracle.get_centroid(cloud, False), oracle.get_mean_distance(cloud, False), oracle.pca_axis(cloud.T)
            c_g, md_g, x0_g = K.centroid(x).cpu().numpy(), K.mean_distance(x).item(), K.pca_axis(x).cpu().numpy()
            same_stats = np.array_equal(c_g, np.ravel(c_o)) and (md_g == md_o or (np.isnan(md_g) and np.isnan(md_o)))
            axis_err = float(min(np.abs(x0_g - x0_o).max(), np.abs(x0_g + x0_o).max()))
            if not same_stats:
                fails.append(tag + ": statistics differ (%r vs %r)" % (md_g, md_o))
            x0_p = SC.pca_axis_host(SC.pca_view(cloud))          # the product's axis (sklearn's computation on the caller's array)
            if not (np.array_equal(x0_p, x0_o) or (np.isnan(x0_p).all() and np.isnan(x0_o).all())):
                fails.append(tag + ": PCA axis of the product differs from the oracle's by %.1e" % float(np.abs(x0_p - x0_o).max()))
            # the axis of a (nearly) isotropic or degenerate cloud is ill-conditioned in any arithmetic: histograms are compared on
            # the ORACLE's axis, handed to both sides, so that a stage is judged on its own inputs
            counts["statistics"] += 1
            oc, ot = oracle.shape_context_counts(c_o, md_o, cloud, typ, x0=x0_o)
            for path in ("tiled", "general"):
                r = K.shape_context(x, d(np.ravel(c_o)), d(x0_o), d(np.array([md_o])), nf, want_counts=True, want_hist=True, path=path)
                if not (np.array_equal(r["counts"].cpu().numpy(), oc) and np.array_equal(r["totals"].cpu().numpy(), ot)):
                    fails.append(tag + ": %s histograms differ (%s path), axis error %.1e" % (typ, path, axis_err))
                if path == "tiled" and r.get("guard") is not None:
                    counts["guard_hits"] += int(np.asarray(r["guard"].cpu()).sum())
            desc[typ] = oracle.normalise_counts(oc, ot)
            counts["descriptors"] += 1
        um, uf = desc["moving"], desc["fixed"]
        U8 = K.chi2_cost8(d(um), d(uf)).cpu().numpy()
        for h, nm in enumerate(oracle.HYPOTHESES):
            want = oracle.unary_distance_matrix(um[int(nm[0]) - 1], uf[int(nm[1]) - 1])
            if not np.array_equal(U8[h].view(np.uint64), np.asarray(want).view(np.uint64)):
                both_nan = np.isnan(U8[h]) & np.isnan(want)
                if not np.array_equal(np.where(both_nan, 0.0, U8[h]).view(np.uint64), np.where(both_nan, 0.0, want).view(np.uint64)):
                    fails.append(tag + ": cost matrix %s differs" % nm)
        counts["costs"] += 1
        # the complete registration (seeded: NumPy's RANSAC stream on both sides)
        det, odet = {}, {}
        ref = oracle.estimate_transform(mv, fx, transform=transform, ransac_trials=80, ransac_error=25.0 * (np.abs(mv).max() / 300.0 + 1e-9),
                                        icp_iterations=4, seed=rs, details=odet)
        got = platymatch_amd.register(mv, fx, transform=transform, ransac_trials=80, ransac_error=25.0 * (np.abs(mv).max() / 300.0 + 1e-9),
                                      icp_iterations=4, seed=rs, details=det)
        g = det.get("edge_guard") or {}
        gsum = sum(v for side in g.values() for v in side.values())
        for h in range(8):
            if not (np.array_equal(det["lsa"][h][0], odet["lsa"][h][0]) and np.array_equal(det["lsa"][h][1], odet["lsa"][h][1])):
                fails.append(tag + ": assignment of hypothesis %d differs (edge guard of the call: %d)" % (h, gsum))
                if gsum == 0:
                    counts["unguarded_assignment_mismatch"] = counts.get("unguarded_assignment_mismatch", 0) + 1
                break
        if not np.array_equal(got[2], ref[2]):
            fails.append(tag + ": inlier counts differ %s vs %s" % (list(got[2]), list(ref[2])))
        well = np.isfinite(ref[0]).all() and np.isfinite(ref[1]).all() and np.linalg.cond(ref[0]) < 1e8
        if well:
            if not np.array_equal(det["nn"], odet["nn"]):
                fails.append(tag + ": ICP correspondences differ")
            if not (relerr(got[0], ref[0]) < 1e-6 and relerr(got[1] @ got[0], ref[1] @ ref[0]) < 1e-6):
                fails.append(tag + ": transforms differ (%.1e, %.1e)" % (relerr(got[0], ref[0]), relerr(got[1] @ got[0], ref[1] @ ref[0])))
        else:
            counts["skipped_registration"] += 1
        counts["registrations"] += 1
        counts["similar"] += transform == "Similar"
    except Exception as e:                                   # a refusal (ties beyond the solver, degenerate input) must be the SAME on both sides
        try:
            oracle.estimate_transform(mv, fx, transform=transform, ransac_trials=80, ransac_error=25.0 * (np.abs(mv).max() / 300.0 + 1e-9),
                                      icp_iterations=4, seed=rs)
            fails.append(tag + ": the product raised %s: %s, the oracle did not" % (type(e).__name__, str(e)[:200]))
        except Exception:
            counts["skipped_registration"] += 1
    counts["cases"] += 1
    counts["lattice"] += lattice
    seed += 1
    if time.perf_counter() > t_print:
        print("... %d cases, %d mismatches" % (counts["cases"], len(fails)), flush=True)
        t_print = time.perf_counter() + 30.0
print("soak: seeds %d..%d, up to %d points: %s" % (seed0, seed - 1, max_points, counts))
print("mismatches: %d" % len(fails))
for f in fails[:40]:
    print("  " + f)
sys.exit(1 if fails else 0)
