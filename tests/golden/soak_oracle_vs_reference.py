#!/usr/bin/env python3
"""The CPU oracle against the UNMODIFIED reference on a stream of small random inputs — what pins the oracle beyond the committed
fixtures.  Runs only in the build container (the reference is mounted at /root/reference; same import recipe as gen_golden.py);
nothing here is used at test time.  Per case: get_centroid / get_mean_distance (bits), get_unary's histograms (as counts) for
generic, lattice and planar clouds, get_unary_distance on them (bits), seeded do_ransac (inliers, model), perform_icp (4x4),
get_affine_transform / get_similar_transform / apply_affine_transform (bits where both are NumPy calls).
Usage: python tests/golden/soak_oracle_vs_reference.py [seconds] [max_points] [first_seed]"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from gen_golden import import_reference  # noqa: E402
import oracle  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
max_points = int(sys.argv[2]) if len(sys.argv) > 2 else 40
seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 0
sc, ft, at, icp, ut = import_reference()
oracle.build()


def relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


R_EDGES = np.logspace(np.log10(1 / 8), np.log10(2), 5)


def guard_reach(cloud):
    """Per queried row: would the product's edge guard (csrc/pm_shape_context.hip: PM_GUARD_REF) count one of its neighbours?
    Restated in NumPy from the same geometry: a neighbour within 1.6e-13 x |d|_1 of a ring radius, a sector plane or a polar cone."""
    c = np.asarray(oracle.get_centroid(cloud, False)).ravel()
    md = oracle.get_mean_distance(cloud, False)
    x0 = oracle.pca_axis(cloud.T)
    n = cloud.shape[1]
    hit = np.zeros(n, dtype=bool)
    for i in range(n):
        d = cloud[:, i]
        z = (d - c) / np.linalg.norm(d - c)
        x = x0 - z * np.dot(x0, z)
        x /= np.linalg.norm(x)
        y = np.cross(z, x)
        v = np.delete(cloud, i, 1) - d[:, None]
        lx, ly, lz = np.abs(x @ v), np.abs(y @ v), np.abs(z @ v)
        r = np.sqrt(lx * lx + ly * ly + lz * lz)
        pl = np.sqrt(lx * lx + ly * ly)
        noise = 1.6e-13 * np.abs(d).sum()
        ring = np.min(np.abs(r[:, None] - md * R_EDGES[None, :4]), axis=1)
        sector = np.minimum(np.minimum(lx, ly), np.minimum(np.abs(ly - np.tan(np.pi / 6) * lx) * np.cos(np.pi / 6), np.abs(ly - np.tan(np.pi / 3) * lx) * 0.5))
        cone = np.minimum(lz, np.minimum(np.abs(lz * 0.5 - pl * np.cos(np.pi / 6)), np.abs(lz * np.cos(np.pi / 6) - pl * 0.5)))
        hit[i] = bool(np.any((r == 0) | (ring <= noise) | (sector <= noise) | (cone <= noise)))    # r == 0: a duplicate of the query
    return hit


fails = []
kinds = {}
guarded = dict(rows_differ=0, rows_differ_flagged=0, rows_agree=0, rows_agree_flagged=0)
counts = dict(cases=0, hist_rows=0, hist_rows_differ=0, cost_entries=0)
t_end = time.perf_counter() + budget
seed = seed0
while time.perf_counter() < t_end:
    rng = np.random.default_rng(424243 * seed + 7)
    n, m = int(rng.integers(5, max_points + 1)), int(rng.integers(5, max_points + 1))
    kind = ["generic", "lattice", "voxel", "planar"][seed % 4]
    base = rng.normal(size=(3, max(n, m))) * rng.uniform(5, 40, size=(3, 1)) + rng.uniform(0, 200, size=(3, 1))
    A = np.eye(3) + rng.normal(scale=0.1, size=(3, 3))
    mv, fx = base.copy(), A @ base + rng.uniform(-20, 20, size=(3, 1)) + rng.normal(scale=0.5, size=base.shape)
    if kind == "lattice":
        mv, fx = np.round(mv * 0.2) * 5.0, np.round(fx * 0.2) * 5.0
    elif kind == "voxel":
        mv, fx = np.round(mv), np.round(fx)
    elif kind == "planar":
        mv[2], fx[2] = mv[2, 0], fx[2, 0]
    mv, fx = np.ascontiguousarray(mv[:, :n]), np.ascontiguousarray(fx[:, rng.permutation(max(n, m))[:m]])
    tag = "seed %d (%s, N=%d, M=%d)" % (seed, kind, n, m)
    kinds[kind] = kinds.get(kind, 0) + 1
    try:
        # statistics
        for cloud in (mv, fx):
            if not np.array_equal(np.asarray(ut.get_centroid(cloud, transposed=False)), np.asarray(oracle.get_centroid(cloud, False))):
                fails.append(tag + ": get_centroid bits")
            if ut.get_mean_distance(cloud, transposed=False) != oracle.get_mean_distance(cloud, False):
                fails.append(tag + ": get_mean_distance bits")
            ct = np.ascontiguousarray(cloud.T)                                   # the N x 3 layout (transposed=True)
            if not np.array_equal(np.asarray(ut.get_centroid(ct, transposed=True)), np.asarray(oracle.get_centroid(ct, True))):
                fails.append(tag + ": get_centroid(transposed=True) bits")
            if ut.get_mean_distance(ct, transposed=True) != oracle.get_mean_distance(ct, True):
                fails.append(tag + ": get_mean_distance(transposed=True) bits")
        pq = min(n, m)
        if ut.get_error(mv[:, :pq], fx[:, :pq]) != oracle.get_error(mv[:, :pq], fx[:, :pq]):
            fails.append(tag + ": get_error bits")
        # descriptors: the reference's normalised histograms back to counts (x (N-1), exact for these sizes)
        desc_r, desc_o = {}, {}
        for cloud, typ, nf in ((mv, "moving", 2), (fx, "fixed", 4)):
            c, md = ut.get_centroid(cloud, transposed=False), ut.get_mean_distance(cloud, transposed=False)
            got = sc.get_unary(c, md, cloud, type=typ, transposed=False)[:nf]
            oc, ot = oracle.shape_context_counts(oracle.get_centroid(cloud, False), oracle.get_mean_distance(cloud, False), cloud, typ)
            want = oracle.normalise_counts(oc, ot)
            desc_r[typ], desc_o[typ] = np.asarray(got), want
            any_bad = np.zeros(cloud.shape[1], dtype=bool)
            for f in range(nf):
                any_bad |= ~np.all((np.asarray(got[f]) == want[f]) | (np.isnan(np.asarray(got[f])) & np.isnan(want[f])), axis=1)
            if kind != "planar":
                reach = guard_reach(cloud)
                guarded["rows_differ"] += int(any_bad.sum()); guarded["rows_differ_flagged"] += int((any_bad & reach).sum())
                guarded["rows_agree"] += int((~any_bad).sum()); guarded["rows_agree_flagged"] += int((~any_bad & reach).sum())
            for f in range(nf):
                rows_bad = np.flatnonzero(~np.all((np.asarray(got[f]) == want[f]) | (np.isnan(np.asarray(got[f])) & np.isnan(want[f])), axis=1))
                counts["hist_rows"] += cloud.shape[1]
                counts["hist_rows_differ"] += rows_bad.size
                if rows_bad.size:
                    fails.append(tag + ": %s frame %d histograms differ in %d of %d rows" % (typ, f + 1, rows_bad.size, cloud.shape[1]))
        # chi-square on the REFERENCE's descriptors (so that a histogram difference above does not cascade)
        um, uf = desc_r["moving"], desc_r["fixed"]
        rows = rng.permutation(n)[:6]
        cols = rng.permutation(m)[:6]
        for h, nmh in enumerate(oracle.HYPOTHESES):
            a, b = um[int(nmh[0]) - 1], uf[int(nmh[1]) - 1]
            want = np.asarray(oracle.unary_distance_matrix(a[rows], b[cols]))
            for i in range(len(rows)):
                for j in range(len(cols)):
                    r = sc.get_unary_distance(a[rows[i]], b[cols[j]])
                    counts["cost_entries"] += 1
                    if not (r == want[i, j] or (np.isnan(r) and np.isnan(want[i, j]))):
                        fails.append(tag + ": get_unary_distance bits (%r vs %r)" % (r, want[i, j]))
        # RANSAC, ICP, fits on the first min(n, m) pairs
        p = min(n, m)
        rs = int(rng.integers(0, 2 ** 31))
        for transform in ("Affine", "Similar"):
            np.random.seed(rs)
            A_r, inl_r = sc.do_ransac(mv[:, :p], fx[:, :p], 4, 30, 10.0, transform)
            np.random.seed(rs)
            A_o, inl_o = oracle.do_ransac(mv[:, :p], fx[:, :p], 4, 30, 10.0, transform)
            if int(inl_r) != int(inl_o):
                fails.append(tag + ": do_ransac %s inliers %d vs %d" % (transform, inl_o, inl_r))
            elif np.isfinite(A_r).all() and not np.array_equal(np.asarray(A_r), np.asarray(A_o)):
                fails.append(tag + ": do_ransac %s model bits differ (%.1e)" % (transform, relerr(A_o, A_r)))
            if kind != "planar":
                import contextlib, io
                I_r = I_o = e_r = e_o = None
                try:
                    with contextlib.redirect_stdout(io.StringIO()):
                        I_r = icp.perform_icp(mv.copy(), fx.copy(), 5, transform)
                except Exception as e:
                    e_r = e
                try:
                    I_o = oracle.perform_icp(mv, fx, 5, transform)
                except Exception as e:
                    e_o = e
                if e_r is not None or e_o is not None:
                    if e_r is None or e_o is None or type(e_r) is not type(e_o) or str(e_r) != str(e_o):
                        fails.append(tag + ": perform_icp %s: reference %r, oracle %r" % (transform, e_r, e_o))
                    else:
                        counts["icp_both_raised"] = counts.get("icp_both_raised", 0) + 1
                elif np.isfinite(I_r).all() and not np.array_equal(np.asarray(I_r), np.asarray(I_o)):
                    fails.append(tag + ": perform_icp %s bits differ (%.1e)" % (transform, relerr(I_o, I_r)))
        P, Y = mv[:, :p], fx[:, :p]
        if not np.array_equal(ft.get_affine_transform(P, Y), oracle.get_affine_transform(P, Y)):
            fails.append(tag + ": get_affine_transform bits")
        X_r = at.apply_affine_transform(mv, ft.get_affine_transform(P, Y))
        if not np.array_equal(X_r, oracle.apply_affine_transform(mv, oracle.get_affine_transform(P, Y))):
            fails.append(tag + ": apply_affine_transform bits")
    except Exception as e:
        fails.append(tag + ": probe error %r" % (e,))
    counts["cases"] += 1
    seed += 1
print("oracle vs reference: seeds %d..%d, up to %d points, kinds %s: %s" % (seed0, seed - 1, max_points, kinds, counts))
print("mismatches: %d" % len(fails))
print("edge guard (restated; generic + lattice + voxel clouds): rows whose histograms differ from the reference's: %d, of which within the "
      "guard's reach: %d; rows that agree: %d, of which flagged all the same: %d" % (guarded["rows_differ"], guarded["rows_differ_flagged"],
                                                                                      guarded["rows_agree"], guarded["rows_agree_flagged"]))
import re
cats = {}
for f in fails:
    m = re.match(r"seed \d+ \((\w+),[^)]*\): (.*)", f)
    what = re.sub(r"frame \d", "frame k", re.sub(r" in \d+ of \d+ rows| \(.*\)$|\d+ vs \d+", "", m.group(2)))
    cats[(m.group(1), what)] = cats.get((m.group(1), what), 0) + 1
for (kind, what), c in sorted(cats.items(), key=lambda kv: -kv[1]):
    print("%6d  %-8s %s" % (c, kind, what))
cases_with = {}
for f in fails:
    m = re.match(r"seed (\d+) \((\w+),[^)]*\): (\w+) frame", f)
    if m:
        cases_with.setdefault(m.group(2), set()).add(m.group(1))
print("cases with any histogram row differing:", {k: "%d of %d" % (len(v), kinds[k]) for k, v in cases_with.items()})
for f in fails[:int(os.environ.get("PM_SOAK_SHOW", "60"))]:
    print("  " + f)
sys.exit(1 if fails else 0)
