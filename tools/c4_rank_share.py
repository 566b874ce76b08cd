#!/usr/bin/env python3
"""One rank's share of BASELINE config 4 (200k x 200k chi-square over 8 GPUs) on one GPU: descriptors of both 200 000-point
clouds' local rows (25 000 each; here the fixed descriptors of all 200 000 rows are built locally instead of gathered), the
25 000 x 200 000 x 8 cost rows in slabs, arg-min per row.  Usage: python tools/c4_rank_share.py [rows_per_slab]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from platymatch_amd import _kernels as K, _native as nat  # noqa: E402

n, ranks, rank = 200_000, 8, 3
slab = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rows = n // ranks
r0 = rank * rows
dev = torch.device("cuda:0")
mv, fx, _ = bench.synth(n, seed=4)
mov, fix = nat.to_dev(mv, dev=dev), nat.to_dev(fx, dev=dev)


def mark():
    torch.cuda.synchronize()
    return time.perf_counter()


K.centroid(mov)
t0 = mark()
sm = (K.centroid(mov), K.pca_axis(mov), K.mean_distance(mov))
sf = (K.centroid(fix), K.pca_axis(fix), K.mean_distance(fix))
t1 = mark()
hm = K.shape_context(mov, *sm, 2, row0=r0, nrows=rows)["hist"]
hf_local = K.shape_context(fix, *sf, 4, row0=r0, nrows=rows)["hist"]          # what this rank would contribute
t2 = mark()
hf = K.shape_context(fix, *sf, 4)["hist"]                                      # stands in for the all-gather
assert K.chi2_symmetric(hm, hf)
t3 = mark()
buf = torch.empty((8, slab, n), dtype=torch.float64, device=dev)
idx = torch.empty((8, rows), dtype=torch.int32, device=dev)
for s0 in range(0, rows, slab):
    s1 = min(rows, s0 + slab)
    U = K.chi2_cost8_frame1(hm[0, s0:s1].contiguous(), hf[0], out=buf[:, :s1 - s0])
    idx[:, s0:s1] = K.row_argmin(U)
t4 = mark()
pairs = rows * n
print("statistics (replicated, 200k)         %8.1f ms" % ((t1 - t0) * 1e3))
print("descriptors, this rank's 2 x 25k rows %8.1f ms" % ((t2 - t1) * 1e3))
print("(all 200k fixed descriptors, stand-in for the gather: %.1f ms)" % ((t3 - t2) * 1e3))
print("8 x 25k x 200k cost rows + arg-min     %8.1f ms   %.3e point-pairs/s per rank, slab %d rows (%.1f GB)"
      % ((t4 - t3) * 1e3, pairs / (t4 - t3), slab, buf.numel() * 8 / 1e9))
print("arg-min histogram check: %d distinct targets in U11" % int(torch.unique(idx[0]).numel()))
