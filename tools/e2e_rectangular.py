#!/usr/bin/env python3
"""A complete registration of clouds of DIFFERENT sizes (the rule for real specimen pairs), wall clock with the stage split.
Usage: python tools/e2e_rectangular.py N M   (N moving, M fixed nuclei; the larger cloud is generated and the other cut from it)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import synth_pair  # noqa: E402
from platymatch_amd import pipeline as P  # noqa: E402
from platymatch_amd.estimate_transform import perform_icp as pi  # noqa: E402

n, m = int(sys.argv[1]), int(sys.argv[2])
pi.VERBOSE = False
mv, fx, A_gt = synth_pair(max(n, m), 42)
rng = np.random.default_rng(7)
mv = np.ascontiguousarray(mv[:, np.sort(rng.choice(mv.shape[1], n, replace=False))])
fx = np.ascontiguousarray(fx[:, np.sort(rng.choice(fx.shape[1], m, replace=False))])
be = P.GpuBackend()
mov, fix = be.cloud(mv), be.cloud(fx)
for rep in range(int(os.environ.get("PM_E2E_REPEAT", "3"))):
    det = {"timing": True}
    t = time.perf_counter()
    A_sc, A_icp, inl = P.estimate_transform(mov, fix, ransac_trials=8000, ransac_error=16, icp_iterations=50, seed=0, details=det)
    torch.cuda.synchronize()
    final = A_icp.cpu().numpy() @ A_sc.cpu().numpy()
    print("N = %d, M = %d: %.3f s; inliers %s; rel. error vs ground truth %.1e; stages %s; routes %s" % (
        n, m, time.perf_counter() - t, list(inl), np.linalg.norm(final - A_gt) / np.linalg.norm(A_gt),
        {k: round(v, 3) for k, v in det["timing"].items()}, sorted(set(det["assignment"]["routes"]))), flush=True)
