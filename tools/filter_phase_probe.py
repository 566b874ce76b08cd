#!/usr/bin/env python3
"""A complete registration in cost_mode='filter', three calls, and what the filtered solve of each pairing did: auction and search
times, the exact-pricing (polishing) rounds, how many entries were listed and evaluated exactly.  Usage: python tools/filter_phase_probe.py N"""
import sys, time, numpy as np, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import synth_pair
from platymatch_amd import pipeline as P
from platymatch_amd.estimate_transform import perform_icp as pi
pi.VERBOSE=False
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
mv,fx,_=synth_pair(n,42)
for rep in range(3):
    det={"timing":True}
    t=time.perf_counter()
    P.estimate_transform(mv,fx,ransac_trials=8000,ransac_error=16,icp_iterations=50,details=det,cost_mode='filter')
    torch.cuda.synchronize()
    print("wall %.1f ms"%((time.perf_counter()-t)*1e3), {k:round(v,3) for k,v in det["timing"].items()})
for h in range(4):
    d=det["assignment"]["details"][h]
    print(h, {k:(round(d[k],4) if isinstance(d[k],float) else d[k]) for k in ("rounds","steps","auction_bids","auction_seconds","core_seconds","polish_seconds","polish_list_shape","polish_listed","polish_violated","listed","tight_within_eps") if k in d})
