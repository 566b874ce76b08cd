#!/usr/bin/env python3
"""A soak-family-C case under the magnifying glass: perform_icp of the product against the oracle, iteration by iteration
(clouds, fits, conditioning).  Usage: python tests/probes/soak_debug_icp2.py SEED"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import oracle  # noqa: E402
from platymatch_amd import _kernels as K, _native as nat  # noqa: E402
from platymatch_amd.estimate_transform import perform_icp as pi  # noqa: E402

seed = int(sys.argv[1])
oracle.build(); nat.load(); pi.VERBOSE = False
dev = torch.device("cuda:0")
rng = np.random.default_rng(9000011 * seed + 3)
n, m = int(rng.integers(5, 1501)), int(rng.integers(5, 1501))
lattice = seed % 4 == 3
base = rng.normal(size=(3, max(n, m))) * rng.uniform(10, 60, size=(3, 1)) + rng.uniform(-100, 300, size=(3, 1))
th = rng.uniform(-0.08, 0.08)
R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
fx = base + rng.normal(scale=0.6, size=base.shape)
mv = rng.uniform(0.97, 1.03) * (R @ base) + rng.uniform(-3, 3, size=(3, 1))
if lattice:
    mv, fx = np.round(mv * 0.5) * 2.0, np.round(fx * 0.5) * 2.0
mv, fx = np.ascontiguousarray(mv[:, :n]), np.ascontiguousarray(fx[:, rng.permutation(base.shape[1])[:m]])
print("N=%d M=%d lattice=%s" % (n, m, lattice))
glog = {}
A_g = pi.perform_icp(mv, fx, 12, "Affine", log=glog)
A_g = A_g.cpu().numpy() if hasattr(A_g, "cpu") else np.asarray(A_g)
cur = mv.copy()
A_icp = np.eye(4)
st = torch.zeros(1, dtype=torch.int32, device=dev)
work = nat.to_dev(mv, dev=dev).clone()
Ad, resd, nnd = K.icp(work, nat.to_dev(fx, dev=dev), 12, want_nn=True, status=st)
print("device loop status (1 = degenerate somewhere):", int(st.item()))
for it in range(12):
    i2, _ = oracle.nn_argmin(cur, fx)
    A = oracle.get_affine_transform(cur, fx[:, i2])
    c = cur - cur.mean(1, keepdims=True)
    C = c @ c.T
    ratio = np.linalg.det(C) / (np.trace(C) / 3) ** 3
    sv = np.linalg.svd(np.vstack([cur, np.ones((1, n))]), compute_uv=False)
    print("iteration %2d: distinct targets %3d, det/(tr/3)^3 %.2e, singular values %s, nn equal to product's: %s"
          % (it, len(np.unique(i2)), ratio, np.array2string(sv, precision=2), np.array_equal(i2, np.asarray(glog["nn"][it].cpu() if hasattr(glog["nn"][it], "cpu") else glog["nn"][it]))))
    cur = oracle.apply_affine_transform(cur, A)
    A_icp = A @ A_icp
print("final relerr product vs oracle: %.2e; device-loop A vs oracle: %.2e" % (np.linalg.norm(A_g - A_icp) / np.linalg.norm(A_icp),
                                                                           np.linalg.norm(Ad.cpu().numpy() - A_icp) / np.linalg.norm(A_icp)))
