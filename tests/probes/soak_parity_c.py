#!/usr/bin/env python3
"""The soak's third family: the reference's functions one by one through the product's mirror (platymatch_amd.estimate_transform.*,
platymatch_amd.utils.utils) against the CPU oracle on random inputs —
  perform_icp       12 iterations from a random perturbation, Affine and Similar, lattice and generic clouds, up to 1 500 points:
                    every correspondence of every iteration, the residuals and the 4x4
  do_ransac         seeded, 3..8 pairs per sample, both modes: inlier count and model
  get_affine_transform / get_similar_transform / apply_*   on random and degenerate (planar, collinear, repeated) point sets
  get_centroid / get_mean_distance / get_error             both layouts
Usage: python tests/probes/soak_parity_c.py [seconds] [first_seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401
import oracle  # noqa: E402
from platymatch_amd import _native as nat  # noqa: E402
from platymatch_amd.estimate_transform import perform_icp as pi, shape_context as sc, find_transform as ft, apply_transform as at  # noqa: E402
from platymatch_amd.utils import utils as ut  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
oracle.build(); nat.load(); pi.VERBOSE = False


def relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def host(x):
    return x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)


counts = dict(cases=0, icp=0, ransac=0, fits=0, stats=0)
fails = []
t_end = time.perf_counter() + budget
t_print = time.perf_counter() + 30.0
seed = seed0
while time.perf_counter() < t_end:
    rng = np.random.default_rng(9000011 * seed + 3)
    n, m = int(rng.integers(5, 1501)), int(rng.integers(5, 1501))
    lattice = seed % 4 == 3
    base = rng.normal(size=(3, max(n, m))) * rng.uniform(10, 60, size=(3, 1)) + rng.uniform(-100, 300, size=(3, 1))
    th = rng.uniform(-0.08, 0.08)
    R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
    fx = base + rng.normal(scale=0.6, size=base.shape)
    mv = rng.uniform(0.97, 1.03) * (R @ base) + rng.uniform(-3, 3, size=(3, 1))
    if lattice:
        mv, fx = np.round(mv * 0.5) * 2.0, np.round(fx * 0.5) * 2.0
    mv, fx = np.ascontiguousarray(mv[:, :n]), np.ascontiguousarray(fx[:, rng.permutation(base.shape[1])[:m]])
    transform = "Similar" if seed % 3 == 1 else "Affine"
    tag = "seed %d (N=%d, M=%d%s, %s)" % (seed, n, m, ", lattice" if lattice else "", transform)
    # ---- perform_icp
    try:
        olog, glog = {}, {}
        oexc = gexc = None
        try:
            A_o = oracle.perform_icp(mv, fx, 12, transform, log=olog)
        except Exception as e:
            oexc = e
        try:
            A_g = host(pi.perform_icp(mv, fx, 12, transform, log=glog))
        except Exception as e:
            gexc = e
        if (oexc is None) != (gexc is None):
            fails.append(tag + ": perform_icp: oracle %s, product %s" % (repr(oexc)[:80], repr(gexc)[:80]))
        elif oexc is None:
            ok_ref = np.isfinite(A_o).all()
            if ok_ref:
                nn_o, nn_g = np.asarray(olog["nn"]), host(glog["nn"])
                bad = (nn_o != nn_g).sum(1)
                if bad.any():
                    first = int(np.flatnonzero(bad)[0])
                    fails.append(tag + ": perform_icp correspondences differ from iteration %d on (%s per iteration)" % (first, list(bad)))
                elif not (relerr(A_g, A_o) < 1e-8 and np.allclose(host(glog["residuals"]), olog["residuals"], rtol=1e-8, atol=1e-10)):
                    fails.append(tag + ": perform_icp transform / residuals differ (%.1e)" % relerr(A_g, A_o))
            elif np.isfinite(A_g).all():
                fails.append(tag + ": perform_icp: the oracle's result is not finite, the product's is")
        counts["icp"] += 1
    except Exception as e:
        fails.append(tag + ": perform_icp probe error %r" % (e,))
    # ---- do_ransac on matched pairs (the first min(n, m) columns paired as they come: mostly outliers, like a wrong hypothesis)
    try:
        k = int(rng.choice([3, 4, 4, 5, 8]))
        p = min(n, m)
        rs = int(rng.integers(0, 2 ** 31))
        o_exc = g_exc = None
        A_o = inl_o = A_g = inl_g = None
        np.random.seed(rs)
        try:
            A_o, inl_o = oracle.do_ransac(mv[:, :p], fx[:, :p], k, 60, 8.0, transform)
        except ValueError as e:                          # more pairs per sample than there are pairs: np.random.choice refuses
            o_exc = e
        np.random.seed(rs)
        try:
            A_g, inl_g = sc.do_ransac(mv[:, :p], fx[:, :p], k, 60, 8.0, transform)
        except ValueError as e:
            g_exc = e
        if o_exc is not None or g_exc is not None:
            if (o_exc is None) != (g_exc is None) or str(o_exc) != str(g_exc):
                fails.append(tag + ": do_ransac(k=%d, %d pairs): oracle %r, product %r" % (k, p, o_exc, g_exc))
        elif int(inl_o) != int(inl_g):
            fails.append(tag + ": do_ransac(k=%d) inliers %d vs %d" % (k, int(inl_g), int(inl_o)))
        elif np.isfinite(A_o).all() and np.linalg.cond(A_o) < 1e8 and relerr(host(A_g), A_o) > 1e-8:
            fails.append(tag + ": do_ransac(k=%d) model differs (%.1e)" % (k, relerr(host(A_g), A_o)))
        counts["ransac"] += 1
    except Exception as e:
        fails.append(tag + ": do_ransac probe error %r" % (e,))
    # ---- fits and applications on small sets: generic, planar, collinear, repeated
    try:
        q = int(rng.integers(4, 40))
        P = rng.normal(size=(3, q)) * 20.0 + 100.0
        kind = seed % 5
        if kind == 1:
            P[2] = P[2, 0]
        elif kind == 2:
            P = P[:, :1] + np.outer(rng.normal(size=3), rng.normal(size=q))
        elif kind == 3:
            P[:, q // 2:] = P[:, :q - q // 2]
        Y = (R @ P) * 1.1 + 5.0 + rng.normal(scale=0.1, size=P.shape)
        A_o, A_g = oracle.get_affine_transform(P, Y), host(ft.get_affine_transform(P, Y))
        tol = 1e-8 if kind in (0, 4) else 1e-6
        if np.isfinite(A_o).all() and not relerr(A_g, A_o) < tol:
            fails.append(tag + ": get_affine_transform kind %d differs (%.1e)" % (kind, relerr(A_g, A_o)))
        if kind in (0, 4):
            S_o, S_g = oracle.get_similar_transform(P, Y), ft.get_similar_transform(P, Y)
            S_o, S_g = (S_o if isinstance(S_o, (tuple, list)) else (S_o,)), (S_g if isinstance(S_g, (tuple, list)) else (S_g,))
            for a, b in zip(S_g, S_o):
                if not np.allclose(host(a), np.asarray(b), rtol=1e-9, atol=1e-9):
                    fails.append(tag + ": get_similar_transform differs")
                    break
        X_o, X_g = oracle.apply_affine_transform(mv, A_o), host(at.apply_affine_transform(mv, A_o))
        if np.isfinite(A_o).all() and not np.array_equal(X_g, X_o):
            fails.append(tag + ": apply_affine_transform is not np.matmul's bits (max abs diff %.1e)" % np.abs(X_g - X_o).max())
        counts["fits"] += 1
    except Exception as e:
        fails.append(tag + ": fit probe error %r" % (e,))
    # ---- statistics, both layouts
    try:
        for cloud in (mv, fx):
            for transposed in (False, True):
                arg = np.ascontiguousarray(cloud.T) if transposed else cloud
                c_o, c_g = oracle.get_centroid(arg, transposed), host(ut.get_centroid(arg, transposed))
                if not np.array_equal(np.ravel(c_g), np.ravel(c_o)):
                    fails.append(tag + ": get_centroid(transposed=%s) bits differ" % transposed)
                d_o, d_g = oracle.get_mean_distance(arg, transposed), float(host(ut.get_mean_distance(arg, transposed)))
                if not (d_o == d_g):
                    fails.append(tag + ": get_mean_distance(transposed=%s) %r vs %r" % (transposed, d_g, d_o))
        e_o, e_g = oracle.get_error(mv[:, :min(n, m)], fx[:, :min(n, m)]), float(host(ut.get_error(mv[:, :min(n, m)], fx[:, :min(n, m)])))
        if not abs(e_o - e_g) <= 1e-12 * abs(e_o):
            fails.append(tag + ": get_error %r vs %r" % (e_g, e_o))
        counts["stats"] += 1
    except Exception as e:
        fails.append(tag + ": statistics probe error %r" % (e,))
    counts["cases"] += 1
    seed += 1
    if time.perf_counter() > t_print:
        print("... %d cases, %d mismatches" % (counts["cases"], len(fails)), flush=True)
        t_print = time.perf_counter() + 30.0
print("soak C: seeds %d..%d: %s" % (seed0, seed - 1, counts))
print("mismatches: %d" % len(fails))
for f in fails[:60]:
    print("  " + f)
sys.exit(1 if fails else 0)
