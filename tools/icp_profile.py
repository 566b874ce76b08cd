"""Affine ICP at 50k points (the bench's refinement stage): time per iteration of the fused loop, and of its pieces.
Kernel-level profile:  cd /tmp && rocprofv3 --kernel-trace --stats -d out -o icp -- python3 tools/icp_profile.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from platymatch_amd import _kernels as K, _native as nat

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
mv, fx, start = bench.synth(n)
dev = torch.device("cuda:0")
fix, st = nat.to_dev(fx, dev=dev), nat.to_dev(start, dev=dev)
ws = nat.workspace(nat.load().pm_icp_workspace(n, n), dev)
for _ in range(2):
    A, res, _ = K.icp(st.clone(), fix, iters, ws=ws)
torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    w = st.clone()
    e0.record()
    A, res, _ = K.icp(w, fix, iters, ws=ws)
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
print("%d iterations at n=%d: %.2f ms = %.1f us per iteration; residual %.6f -> %.6f" % (iters, n, best, best * 1e3 / iters, float(res[0]), float(res[-1])))
# the pieces, stand-alone: grid search of all points, moment sums, update
grid = K.icp_grid(fix)
origin = torch.cat([fix[:, 0], fix[:, 0]]).contiguous()
A2 = torch.eye(4, dtype=torch.float64, device=dev).reshape(16).contiguous()
w = st.clone()
for name, fn in (("grid search (rings)", lambda: K.icp_nn(w, fix, want_dist=False, grid=grid)),):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("%s: %.1f us per call" % (name, e0.elapsed_time(e1) * 1e3 / 50))
