#!/usr/bin/env python3
"""Wall-clock breakdown of one complete registration (BASELINE config 2 scale) — where the time goes once the O(N*M)
stages run on the GPU.  Usage: python tools/e2e_timing.py [N] [ransac_trials]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import synth_pair  # noqa: E402
from platymatch_amd import pipeline as P  # noqa: E402
from platymatch_amd.estimate_transform import perform_icp as pi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
pi.VERBOSE = False
mv, fx, A_gt = synth_pair(n, 42)
be = P.GpuBackend()
mov, fix = be.cloud(mv), be.cloud(fx)
torch.cuda.synchronize()


def timed(label, fn):
    torch.cuda.synchronize()
    t = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    print("%-34s %9.1f ms" % (label, (time.perf_counter() - t) * 1e3), flush=True)
    return out


timed("warm-up (library load, first launches)", lambda: P.build_costs(be, mov[:, :256].contiguous(), fix[:, :256].contiguous()))
U, bn = timed("descriptors + 8 cost matrices (GPU)", lambda: P.build_costs(be, mov, fix))
lsa = timed("8 x assignment (core on host, matrix on GPU)", lambda: P.assign(U, bn))
np.random.seed(0)
res = timed("8 x RANSAC, %d trials (host RNG + GPU)" % trials,
            lambda: [be.do_ransac(mov, fix, r.astype(np.int32), c.astype(np.int32), trials, 16, "Affine", 4) for r, c in lsa])
inl = [k for _, k in res]
A_sc = res[int(np.argmax(inl))][0]
moved = be.apply_affine(P.nat.to_dev(A_sc, dev=mov.device), mov)
A_icp = timed("ICP, 50 iterations (GPU)", lambda: be.icp(moved, fix, 50, "Affine", None))
final = A_icp.cpu().numpy() @ (A_sc.cpu().numpy() if hasattr(A_sc, "cpu") else A_sc)
print("inliers", inl, " rel. error vs ground truth %.2e" % (np.linalg.norm(final - A_gt) / np.linalg.norm(A_gt)))

del U            # the staged run's eight matrices (64 N M bytes: 160 GB at 50k) must go before the driver builds its own
torch.cuda.empty_cache()
# the same registration through the driver, where the RANSAC index sets are drawn while the solver runs and each
# solver thread fetches its own matrix
# (three times: the first call pays for a fresh allocation of the eight matrices — 22 ms per GB — and the host threads' speed
# drifts with the box's load)
for rep in range(int(os.environ.get("PM_E2E_REPEAT", "3"))):
    det = {"timing": True}
    t = time.perf_counter()
    A_sc2, A_icp2, inl2 = P.estimate_transform(mov, fix, ransac_trials=trials, ransac_error=16, icp_iterations=50, seed=0, details=det)
    torch.cuda.synchronize()
    print("%-34s %9.1f ms   (same inliers: %s)  stages %s" % ("estimate_transform, whole", (time.perf_counter() - t) * 1e3, list(inl2) == inl,
                                                            {k: round(v, 3) for k, v in det.get("timing", {}).items()}), flush=True)

# the default for callers who do not seed (the widget never does): index sets drawn on the device in front of each fit
for rep in range(int(os.environ.get("PM_E2E_REPEAT", "3"))):
    det = {"timing": True}
    t = time.perf_counter()
    A_sc3, A_icp3, inl3 = P.estimate_transform(mov, fix, ransac_trials=trials, ransac_error=16, icp_iterations=50, details=det)
    torch.cuda.synchronize()
    final3 = (A_icp3 @ A_sc3).cpu().numpy()
    print("%-34s %9.1f ms   (inliers %s, rel. error %.2e)  stages %s"
          % ("estimate_transform, unseeded", (time.perf_counter() - t) * 1e3, list(inl3), np.linalg.norm(final3 - A_gt) / np.linalg.norm(A_gt),
             {k: round(v, 3) for k, v in det.get("timing", {}).items()}), flush=True)

# the opt-in modes: 'relaxed' (solved on relaxed float64 matrices, certified on the exact matrices' listed entries) and 'filter' (a
# float32 build only selects entries, every cost is exact, no exact matrix is built)
for mode in ("relaxed", "filter"):
    for rep in range(int(os.environ.get("PM_E2E_REPEAT", "3"))):
        det = {"timing": True}
        t = time.perf_counter()
        A_sc4, A_icp4, inl4 = P.estimate_transform(mov, fix, ransac_trials=trials, ransac_error=16, icp_iterations=50, details=det, cost_mode=mode)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t) * 1e3
        modes = [d.get("cost_mode", "?") for d in det.get("assignment", {}).get("details", [])]
        print("%-34s %9.1f ms   (hypotheses settled without an exact matrix %d of 8, built exactly %d; assignments equal the exact mode's: %s)  stages %s"
              % ("estimate_transform, " + mode, wall, sum(m.startswith(("relaxed", "filter")) for m in modes), sum(m.startswith("exact") for m in modes),
                 all(np.array_equal(x[1], y[1]) for x, y in zip(det["lsa"], lsa)), {k: round(v, 3) for k, v in det.get("timing", {}).items()}), flush=True)
