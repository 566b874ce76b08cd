#!/usr/bin/env python3
"""Assignment stage of one synthetic pair, hypothesis by hypothesis: route taken, pricing rounds, core size, timings.
Usage: python tools/lsap_probe.py N SEED [dense]   (dense: also time pm_lsap_solve on hypothesis 0 and 1)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import synth_pair  # noqa: E402
from platymatch_amd import lsap as L, pipeline as P  # noqa: E402

n, seed = int(sys.argv[1]), int(sys.argv[2])
mv, fx, _ = synth_pair(n, seed)
if os.environ.get("PM_LSAP_M"):                      # a rectangular problem: fewer fixed points
    fx = np.ascontiguousarray(fx[:, :int(os.environ["PM_LSAP_M"])])
if os.environ.get("PM_LSAP_N"):
    mv = np.ascontiguousarray(mv[:, :int(os.environ["PM_LSAP_N"])])
be = P.GpuBackend()
U, _ = P.build_costs(be, be.cloud(mv), be.cloud(fx))
torch.cuda.synchronize()
if os.environ.get("PM_LSAP_AUCTION") == "0":
    L.AUCTION = None
for key in ("bids_per_row", "eps0", "eps_min", "factor", "rounds", "later_eps0", "stop_below"):      # e.g. PM_LSAP_bids_per_row=100
    if L.AUCTION is not None and os.environ.get("PM_LSAP_" + key):
        L.AUCTION[key] = type(L.AUCTION[key])(float(os.environ["PM_LSAP_" + key]))
hyps = [int(x) for x in os.environ.get("PM_LSAP_HYPS", "0,1,2,3,4,5,6,7").split(",")]
for h in hyps:
    info = {}
    t = time.perf_counter()
    W = L.DeviceMatrix(U[h] if U.shape[1] <= U.shape[2] else L.transposed(U[h]))       # rows are the short side
    sol = L.solve_core(W, info)
    t1 = time.perf_counter() - t
    t = time.perf_counter()
    ok = sol is not None and L.certify(W, *sol, info=info)
    t2 = time.perf_counter() - t
    print("hyp %d: core %.3f s, certify %.3f s, certified %s, %s" % (h, t1, t2, ok, {k: v for k, v in info.items() if k != "violated_per_round"}), flush=True)
    print("    violated per round:", info.get("violated_per_round"), flush=True)
t = time.perf_counter()
info = {}
L.solve_eight_on_device(U, info=info)
print("solve_eight_on_device: %.3f s, routes %s" % (time.perf_counter() - t, info["routes"]), flush=True)
if len(sys.argv) > 3:
    for h in (0, 1):
        t = time.perf_counter()
        L.linear_sum_assignment(U[h].cpu().numpy())
        print("dense host solver, hyp %d: %.2f s" % (h, time.perf_counter() - t), flush=True)
