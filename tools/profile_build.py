#!/usr/bin/env python3
"""One pass of the 50k descriptor + chi-square cost build (no ICP) for rocprofv3 counter collection.
Usage: rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d <dir> -- python3 tools/profile_build.py [N] [icp_iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from platymatch_amd import _kernels as K, _native as nat, pipeline as P  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda:0")
mv_h, fx_h, start_h = bench.synth(n)
mov, fix, start = (nat.to_dev(x, dev=dev) for x in (mv_h, fx_h, start_h))
be = P.GpuBackend(dev)
cm, mdm, x0m = be.stats(mov)
cf, mdf, x0f = be.stats(fix)
sc_m = be.shape_context(mov, cm, mdm, x0m, 2, 0, n)
sc_f = be.shape_context(fix, cf, mdf, x0f, 4, 0, n)
U = torch.empty((8, n, n), dtype=torch.float64, device=dev)
K.chi2_cost8(sc_m, sc_f, out=U)
if iters:
    K.icp(start.clone(), fix, iters)
torch.cuda.synchronize()
print("done", float(U[0, 0, 0]))
