"""Where the host assignment stage of a 5k registration spends its time: device-to-host copy and solve, per hypothesis,
one after the other (no threads), then the threaded stage as the driver runs it.  Usage: python tools/lsa_breakdown.py [N]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import synth_pair  # noqa: E402
from platymatch_amd import pipeline as P, lsap  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
mv, fx, _ = synth_pair(n, 42)
be = P.GpuBackend()
mov, fix = be.cloud(mv), be.cloud(fx)
U, bn = P.build_costs(be, mov, fix)
torch.cuda.synchronize()
print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "n/a",
      " affinity:", len(os.sched_getaffinity(0)))
for h in range(8):
    t0 = time.perf_counter()
    M = U[h].cpu().numpy()
    t1 = time.perf_counter()
    r, c = lsap.linear_sum_assignment(M)
    t2 = time.perf_counter()
    print("hypothesis %d: copy %6.1f ms  solve %7.1f ms  (identity matches: %d)" % (h, (t1 - t0) * 1e3, (t2 - t1) * 1e3, int((r == c).sum())))
t0 = time.perf_counter()
P.assign(U, bn)
print("threaded stage (8 solves in flight): %.1f ms" % ((time.perf_counter() - t0) * 1e3))
