#!/usr/bin/env python3
"""One soak case's ICP stage under the magnifying glass: product loop against the oracle's, iteration by iteration.
Usage: python tests/probes/soak_debug_icp.py SEED [max_points]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import oracle  # noqa: E402
import platymatch_amd  # noqa: E402
from platymatch_amd import _kernels as K, _native as nat  # noqa: E402
from platymatch_amd.estimate_transform import perform_icp as pi  # noqa: E402

seed = int(sys.argv[1])
max_points = int(sys.argv[2]) if len(sys.argv) > 2 else 500
from soak_cases import make_case  # noqa: E402
mv, fx, lattice, transform, rs = make_case(seed, max_points)
oracle.build(); nat.load(); pi.VERBOSE = False
dev = torch.device("cuda:0")
err = 25.0 * (np.abs(mv).max() / 300.0 + 1e-9)
det, odet = {}, {}
ref = oracle.estimate_transform(mv, fx, transform=transform, ransac_trials=80, ransac_error=err, icp_iterations=4, seed=rs, details=odet)
got = platymatch_amd.register(mv, fx, transform=transform, ransac_trials=80, ransac_error=err, icp_iterations=4, seed=rs, details=det)
print("N=%d M=%d lattice=%s %s; inliers %s / %s" % (mv.shape[1], fx.shape[1], lattice, transform, list(got[2]), list(ref[2])))
print("A_sc equal bits:", np.array_equal(got[0], ref[0]), " max abs diff", np.abs(got[0] - ref[0]).max())
print("cond(A_sc) %.3e" % np.linalg.cond(ref[0]))
nn_g, nn_o = np.asarray(det["nn"]), np.asarray(odet["nn"])
print("nn shapes", nn_g.shape, nn_o.shape)
# replay the oracle's loop from the ORACLE's A_sc, and the product's kernels from the same start, iteration by iteration
moved = oracle.apply_affine_transform(mv, ref[0])
cur_o = moved.copy()
for it in range(nn_o.shape[0]):
    i_o, d_o = oracle.nn_argmin(cur_o, fx)
    i_g = K.icp_nn(nat.to_dev(cur_o, dev=dev), nat.to_dev(fx, dev=dev))[0].cpu().numpy() if hasattr(K, "icp_nn") else None
    bad = np.flatnonzero(nn_g[it] != nn_o[it])
    print("iteration %d: loop nn differs at %d points; stand-alone grid query on the oracle's cloud differs at %s points"
          % (it, bad.size, "n/a" if i_g is None else int((i_g != i_o).sum())))
    for p in bad[:5]:
        a, b = int(nn_g[it][p]), int(nn_o[it][p])
        da = np.sqrt(((fx[:, a] - cur_o[:, p]) ** 2).sum()); db = np.sqrt(((fx[:, b] - cur_o[:, p]) ** 2).sum())
        print("    point %d: product -> %d (dist %.17g), oracle -> %d (dist %.17g); same coordinates: %s" % (p, a, da, b, db, np.array_equal(fx[:, a], fx[:, b])))
    A_est = oracle.get_affine_transform(cur_o, fx[:, i_o]) if transform == "Affine" else oracle.get_similar_transform(cur_o, fx[:, i_o])
    print("    oracle A_est cond %.3e" % np.linalg.cond(A_est))
    cur_o = oracle.apply_affine_transform(cur_o, A_est)
# the product's own loop, stand-alone, from the oracle's start
A, res, nn_all = K.icp(nat.to_dev(moved, dev=dev), nat.to_dev(fx, dev=dev), nn_o.shape[0], want_nn=True) if "want_nn" in K.icp.__code__.co_varnames else (None, None, None)
if nn_all is not None:
    nn_all = nn_all.cpu().numpy()
    for it in range(nn_o.shape[0]):
        print("stand-alone K.icp from the oracle's start, iteration %d: differs at %d points" % (it, int((nn_all[it] != nn_o[it]).sum())))
print("final: A_icp relerr %.3e" % (np.linalg.norm(got[1] - ref[1]) / np.linalg.norm(ref[1])))
print("product A_icp\n", got[1], "\noracle A_icp\n", ref[1])
