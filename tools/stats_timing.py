#!/usr/bin/env python3
"""Launch time of the mean pairwise distance (pm_mean_distance: mean_distance_chunks + the serial finish) and of the whole
statistics stage on the bench's 50 000-point clouds, by HIP events on the launching stream.
Usage: python tools/stats_timing.py [points] [repeats]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from platymatch_amd import _kernels as K, _native as nat, pipeline as P  # noqa: E402
from platymatch_amd.estimate_transform.shape_context import pca_view  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
nat.load()
dev = torch.device("cuda:0")
mv_h, fx_h, _ = bench.synth(n)
mov, fix = nat.to_dev(mv_h, dev=dev), nat.to_dev(fx_h, dev=dev)
for _ in range(3):
    md = K.mean_distance(mov)
torch.cuda.synchronize()
ts = []
for _ in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    md = K.mean_distance(mov)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
be = P.GpuBackend(dev)
views = (pca_view(mv_h), pca_view(fx_h))
st = []
for _ in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    P.statistics_of_both(be, mov, fix, None, views)
    e1.record()
    torch.cuda.synchronize()
    st.append(e0.elapsed_time(e1))
print("n=%d: mean distance %.3f ms (min of %d; median %.3f), value %s; statistics of both clouds %.3f ms (min; median %.3f)"
      % (n, min(ts), reps, float(np.median(ts)), float(md.item()).hex(), min(st), float(np.median(st))))
