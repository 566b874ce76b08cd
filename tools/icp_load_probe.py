"""Diagnostic: the one-launch ICP loop next to a busy stream — where do its results leave the quiet run's?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from platymatch_amd import _kernels as K, _native as nat

n, iters = 50000, 40
mv, fx, start = bench.synth(n)
dev = torch.device("cuda:0")
fix, st, mov = nat.to_dev(fx, dev=dev), nat.to_dev(start, dev=dev), nat.to_dev(mv, dev=dev)
q = st.clone()
s0 = torch.zeros(1, dtype=torch.int32, device=dev)
A0, res0, nn0 = K.icp(q, fix, iters, want_nn=True, status=s0)
a = torch.rand((1024, 360), dtype=torch.float64, device=dev) + 0.1
b = torch.rand((65536, 360), dtype=torch.float64, device=dev) + 0.1
noise = torch.cuda.Stream()
torch.cuda.synchronize()
for rep in range(8):
    for one in (True, False):
        with torch.cuda.stream(noise):
            for _ in range(3 + rep % 4):
                K.chi2_cost(a, b)
        w = st.clone()
        s = torch.zeros(1, dtype=torch.int32, device=dev)
        A, res, nn = K.icp(w, fix, iters, want_nn=True, status=s, one_launch=one)
        torch.cuda.synchronize()
        dres = (res != res0).cpu().numpy()
        dnn = (nn != nn0).any(1).cpu().numpy()
        print("rep %d one_launch=%s status=%d  A equal %s  first differing residual %s  first differing nn row %s  max|dA| %.3e"
              % (rep, one, int(s.item()), bool(torch.equal(A, A0)), (int(np.argmax(dres)) if dres.any() else None),
                 (int(np.argmax(dnn)) if dnn.any() else None), float((A - A0).abs().max())), flush=True)
