#!/usr/bin/env python3
"""Per-iteration cost of the sharded ICP loop for one rank's share (6 250 of 50 000 moving points against the full fixed
cloud) without the collective — what an 8-GPU run pays per iteration besides the all-gather."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from platymatch_amd import _native as nat, pipeline as P  # noqa: E402

dev = torch.device("cuda:0")
mv, fx, start = bench.synth(50000)
fix, st = nat.to_dev(fx, dev=dev), nat.to_dev(start, dev=dev)
be = P.GpuBackend(dev)
for share in (50000, 25000, 12500, 6250):
    loc = st[:, :share].contiguous()
    P.icp_sharded(be, loc, fix, 5, None)
    torch.cuda.synchronize()
    t = time.perf_counter()
    A, res = P.icp_sharded(be, loc, fix, 200, None)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print("rows %6d: 200 iterations %.1f ms = %.0f us per iteration" % (share, dt * 1e3, dt / 200 * 1e6))
