"""200 affine ICP iterations at 50k points (the bench's refinement stage) for a kernel-level profile:
cd /tmp && rocprofv3 --kernel-trace --stats -d out -o icp -- python3 tools/icp_profile.py; then tools/rocprof_stats.py out/icp_results.db"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from platymatch_amd import _kernels as K, _native as nat

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
mv, fx, start = bench.synth(n)
dev = torch.device("cuda:0")
fix, st = nat.to_dev(fx, dev=dev), nat.to_dev(start, dev=dev)
for _ in range(2):
    A, res, _ = K.icp(st.clone(), fix, 200)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
w = st.clone()
e0.record()
A, res, _ = K.icp(w, fix, 200)
e1.record()
torch.cuda.synchronize()
print("200 iterations: %.2f ms, residual %.6f -> %.6f" % (e0.elapsed_time(e1), float(res[0]), float(res[-1])))
