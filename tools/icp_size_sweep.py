#!/usr/bin/env python3
"""The fused ICP loop (grid search bounded by the previous match, reduction tree across workgroups) against the same loop driven
step by step with the brute-force search, bit for bit, over a sweep of sizes — odd ones, ones around the tree's group / chunk
boundaries (512-point groups; 128 groups per fetch), N != M, up to a million points.
Usage: python tools/icp_size_sweep.py [iters]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from platymatch_amd import _kernels as K, _native as nat  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6
nat.load()
dev = torch.device("cuda:0")
sizes = [(1, 1), (7, 3), (8, 8), (9, 1000), (511, 513), (512, 512), (513, 200), (4095, 4097), (32767, 30000), (32768, 32768), (32769, 40000),
         (65535, 65537), (65536, 65536), (65537, 50000), (100003, 99991), (131073, 131071), (300000, 280000), (1000003, 1000000)]
bad = 0


def bits(a, b):                 # equal bit patterns (a degenerate cloud — one point — fits NaNs on both sides)
    return bool(torch.equal(a.contiguous().view(torch.int64), b.contiguous().view(torch.int64)))


for n, m in sizes:
    rng = np.random.default_rng(n * 7 + m)
    base = rng.normal(size=(3, max(n, m))) * np.array([[60.0], [40.0], [25.0]]) + 200.0
    fx = base[:, :m] + rng.normal(scale=1.0, size=(3, m))
    th = 0.03
    R = np.array([[np.cos(th), -np.sin(th), 0.0], [np.sin(th), np.cos(th), 0.0], [0.0, 0.0, 1.0]])
    c = base.mean(1, keepdims=True)
    st = 1.01 * R @ (base[:, rng.permutation(max(n, m))[:n]] - c) + c + np.array([[2.0], [-1.0], [3.0]])
    fix, start = nat.to_dev(np.ascontiguousarray(fx), dev=dev), nat.to_dev(np.ascontiguousarray(st), dev=dev)
    t0 = time.perf_counter()
    work = start.clone()
    A, res, nn_all = K.icp(work, fix, iters, want_nn=True)
    loc = start.clone()
    A2 = torch.eye(4, dtype=torch.float64, device=dev).reshape(16).contiguous()
    origin = torch.cat([fix[:, 0], fix[:, 0]]).contiguous()
    ok = True
    brute = n * m <= 4e11
    for it in range(iters):
        nn, _ = K.icp_nn(loc, fix, want_dist=False, brute=brute)
        ok &= bool(torch.equal(nn, nn_all[it]))
        sums = K.icp_accumulate(loc, fix, nn, origin, nn_trusted=True)
        _, parts = K.icp_update(sums, origin, loc, fix, nn, A2, nn_trusted=True)
        ok &= bits(parts[0] / parts[1], res[it])
    ok &= bits(A.reshape(16), A2) and bits(work, loc)
    if 2 <= n <= 70000:                       # the one-launch persistent loop (an option): the same bits, or an honest status 2
        w1 = start.clone()
        st = torch.zeros(1, dtype=torch.int32, device=dev)
        A1, res1, nn1 = K.icp(w1, fix, iters, want_nn=True, one_launch=True, status=st)
        if int(st.item()) != 2:
            ok &= bits(A1.reshape(16), A.reshape(16)) and bits(w1, work) and bits(res1, res) and bool(torch.equal(nn1, nn_all))
    torch.cuda.synchronize()
    bad += not ok
    print("N=%8d M=%8d: %s  (%s reference search, %.1f s)" % (n, m, "identical" if ok else "MISMATCH", "brute-force" if brute else "grid (rings)",
                                                           time.perf_counter() - t0), flush=True)
print("mismatching sizes: %d" % bad)
sys.exit(1 if bad else 0)
