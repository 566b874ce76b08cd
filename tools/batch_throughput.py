#!/usr/bin/env python3
"""Aggregate registrations/s of a batch of independent pairs on ONE GPU (BASELINE config 5 is this, times 8 GPUs, at
2k-20k points; "replicas only" -- no collective).  Each pair is a complete unsupervised registration: descriptors, eight
cost matrices, eight Hungarian solves, 8 x 8000 RANSAC trials, 50 ICP iterations.
Usage: python tools/batch_throughput.py [workers] [size size ...]"""
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

pi.VERBOSE = False
workers = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sizes = [int(x) for x in sys.argv[2:]] or [2000, 3000, 2500, 4000, 2000, 5000, 3500, 3000]
pairs, truth = [], []
for k, n in enumerate(sizes):
    mv, fx, A = synth_pair(n, 100 + k)
    pairs.append((mv, fx))
    truth.append(A)
P.estimate_transform(pairs[0][0][:, :300], pairs[0][1][:, :300], ransac_trials=100, icp_iterations=2)      # warm-up
for w in sorted({1, workers}):
    torch.cuda.synchronize()
    t = time.perf_counter()
    out = P.estimate_transform_batch(pairs, workers=w, seeds=list(range(len(pairs))), ransac_trials=8000, ransac_error=16,
                                     icp_iterations=50)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    err = max(np.linalg.norm(o[1] @ o[0] - A) / np.linalg.norm(A) for o, A in zip(out, truth))
    print("workers=%d: %d pairs (sizes %s) in %.2f s -> %.3f registrations/s; worst rel. error vs ground truth %.1e"
          % (w, len(pairs), sizes, dt, len(pairs) / dt, err), flush=True)
