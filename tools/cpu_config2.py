#!/usr/bin/env python3
"""BASELINE config 2 (a 5 000-nucleus pair) on the HOST, complete and TIMED — not extrapolated (VERDICT r04 next #7b): the CPU
restatement of the reference's algorithm (oracle/: its loops in C under OpenMP on every core this process may use, SciPy's
linear_sum_assignment for the eight assignments, one after the other as the widget calls them) runs the whole registration —
statistics, descriptors, eight cost matrices, eight assignments, 8 x 8 000 RANSAC trials, 50 ICP iterations.  Test infrastructure
timed as a baseline; the product never calls it.  The literal reference (pure-Python loops) would need ~9 h for the same pair
(BASELINE.md §2).  Usage: python tools/cpu_config2.py [N] [--json]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(n=5000, trials=8000, icp_iterations=50):
    """-> dict(seconds, stage_s, cores, n, inliers, rel_error): the oracle's complete registration of the synthetic n-point pair."""
    import oracle
    from conftest import synth_pair
    oracle.build()
    mv, fx, A_gt = synth_pair(n, 42)
    cores = oracle.host_threads()
    oracle.set_threads(cores)
    o = oracle
    stage = {}
    clock = [time.perf_counter()]

    def mark(name):
        now = time.perf_counter()
        stage[name] = stage.get(name, 0.0) + now - clock[0]
        clock[0] = now

    t0 = time.perf_counter()
    # the stage order of oracle.estimate_transform (= _dock_widget.py:526-718), with a clock between the stages
    cm, cf = o.get_centroid(mv, transposed=False), o.get_centroid(fx, transposed=False)
    mdm, mdf = o.get_mean_distance(mv, transposed=False), o.get_mean_distance(fx, transposed=False)
    mark("statistics")
    um, uf = o.get_unary(cm, mdm, mv, "moving"), o.get_unary(cf, mdf, fx, "fixed")
    mark("descriptors")
    lsa = []
    for h in o.HYPOTHESES:
        U = o.unary_distance_matrix(um[int(h[0]) - 1], uf[int(h[1]) - 1])
        mark("cost_matrices")
        lsa.append(o.linear_sum_assignment(U))
        mark("assignments_scipy")
    np.random.seed(0)
    inl = np.zeros(8, dtype=np.int64)
    A_h = []
    for k, (r, c) in enumerate(lsa):
        A, inl[k] = o.do_ransac(mv[:, r], fx[:, c], min_samples=4, trials=trials, error=16, transform="Affine")
        A_h.append(A)
    A_sc = A_h[int(np.argmax(inl))]
    mark("ransac")
    A_icp = o.perform_icp(o.apply_affine_transform(mv, A_sc), fx, icp_iterations, "Affine")
    mark("icp")
    seconds = time.perf_counter() - t0
    det = {"timing": stage}
    oracle.set_threads(1)
    err = float(np.linalg.norm(A_icp @ A_sc - A_gt) / np.linalg.norm(A_gt))
    return {"seconds": seconds, "stage_s": {k: round(float(v), 3) for k, v in det.get("timing", {}).items()} if isinstance(det.get("timing"), dict) else None,
            "cores": cores, "n": n, "ransac_trials": trials, "icp_iterations": icp_iterations, "inliers": [int(x) for x in inl], "rel_error_vs_ground_truth": err,
            "what": "oracle.estimate_transform: C restatement of the reference's loops under OpenMP + SciPy's linear_sum_assignment x 8, complete, timed"}


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    out = run(int(args[0]) if args else 5000)
    if "--json" in sys.argv:
        print(json.dumps(out))
    else:
        print("config 2 on the host (%d threads), N = M = %d: %.1f s complete (stages %s); inliers %s; rel. error vs ground truth %.1e"
              % (out["cores"], out["n"], out["seconds"], out["stage_s"], out["inliers"], out["rel_error_vs_ground_truth"]))
