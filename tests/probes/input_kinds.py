#!/usr/bin/env python3
"""One registration, many ways of handing the clouds over — NumPy C / Fortran order, a strided view, nested lists, 4 x N
homogeneous rows, torch CPU and GPU tensors, int64 voxel coordinates: the results must be the same numbers (product only).
Usage: python tests/probes/input_kinds.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import platymatch_amd  # noqa: E402
from platymatch_amd.estimate_transform import perform_icp as pi  # noqa: E402
from soak_cases import make_case  # noqa: E402

pi.VERBOSE = False
bad = 0
for seed, transform in ((3, "Affine"), (8, "Similar"), (11, "Affine")):
    mv, fx, _, _, rs = make_case(seed, 400)
    if seed == 11:
        mv, fx = np.round(mv), np.round(fx)
    kw = dict(transform=transform, ransac_trials=100, ransac_error=20.0, icp_iterations=5, seed=rs)
    ref = platymatch_amd.register(mv, fx, **kw)
    big_m = np.zeros((3, 2 * mv.shape[1])); big_m[:, ::2] = mv
    kinds = {
        "fortran order": (np.asfortranarray(mv), np.asfortranarray(fx)),
        "strided view": (big_m[:, ::2], fx),
        "nested lists": (mv.tolist(), fx.tolist()),
        "4 x N homogeneous": (np.vstack([mv, np.ones((1, mv.shape[1]))]), np.vstack([fx, np.ones((1, fx.shape[1]))])),
        "torch cpu": (torch.as_tensor(mv), torch.as_tensor(fx)),
        "torch gpu": (torch.as_tensor(mv).cuda(), torch.as_tensor(fx).cuda()),
    }
    if seed == 11:
        kinds["int64 voxel coordinates"] = (mv.astype(np.int64), fx.astype(np.int64))
    for name, (a, b) in kinds.items():
        try:
            got = platymatch_amd.register(a, b, **kw)
            same = all(np.array_equal(np.asarray(g.cpu() if hasattr(g, "cpu") else g), np.asarray(r), equal_nan=True) for g, r in zip(got, ref))
            note = "" if same else "   <-- differs"
        except Exception as e:
            same, note = False, "   <-- raised %r" % (e,)
        bad += not same
        print("seed %d %-8s %-26s %s%s" % (seed, transform, name, "same" if same else "DIFFERENT", note))
print("differences: %d" % bad)
sys.exit(1 if bad else 0)
