#!/usr/bin/env python3
"""One complete unsupervised registration of two N-point clouds through the driver, with the stage split — for sizes
at which the eight cost matrices (64 N M bytes) exceed HBM and the pipeline streams the hypotheses two matrices at a time
(N > ~67 000 on one MI355X).  Since round 4 a hypothesis whose optimum has a near-tie is settled on its block (lsap.resolve_near_ties):
the call passes NO accept_near_ties and must not raise.  Usage: python tools/big_registration.py N [ransac_trials [icp_iterations [cost_mode]]]"""
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import synth_pair  # noqa: E402
from platymatch_amd import pipeline as P  # noqa: E402
from platymatch_amd.estimate_transform import perform_icp as pi  # noqa: E402

n = int(sys.argv[1])
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 50
cost_mode = sys.argv[4] if len(sys.argv) > 4 else "auto"
pi.VERBOSE = False
t_start = time.perf_counter()
stop = threading.Event()


def heartbeat():
    while not stop.wait(30.0):
        free, total = torch.cuda.mem_get_info()
        print("[%.0f s] running; HBM in use %.0f GB" % (time.perf_counter() - t_start, (total - free) / 1e9), flush=True)


threading.Thread(target=heartbeat, daemon=True).start()
if os.environ.get("PM_BIG_STACKS"):              # diagnosis: every thread's Python stack every PM_BIG_STACKS seconds
    import faulthandler
    faulthandler.dump_traceback_later(float(os.environ["PM_BIG_STACKS"]), repeat=True)
mv, fx, A_gt = synth_pair(n, 42)
P.estimate_transform(mv[:, :400], fx[:, :400], ransac_trials=50, icp_iterations=2)          # warm-up
det = {"timing": True}
t = time.perf_counter()
A_sc, A_icp, inl = P.estimate_transform(mv, fx, ransac_trials=trials, ransac_error=16, icp_iterations=iters, seed=0, details=det, cost_mode=cost_mode)
torch.cuda.synchronize()
dt = time.perf_counter() - t
stop.set()
err = np.linalg.norm(A_icp @ A_sc - A_gt) / np.linalg.norm(A_gt)
print("N = M = %d, cost_mode=%r: complete registration in %.1f s; inliers %s; rel. error vs ground truth %.2e" % (n, cost_mode, dt, inl.tolist(), err))
print("stage seconds:", {k: round(v, 2) for k, v in det["timing"].items()})
print("assignment:", det["assignment"].get("mode", "eight matrices resident"), det["assignment"].get("routes"))
