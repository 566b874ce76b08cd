#!/usr/bin/env python3
"""Rehearsal of the sharded registration driver with the real GPU backend: G ranks share ONE device (gloo carries the
collectives), every rank runs pipeline.estimate_transform(group=WORLD) and the result is compared with the one-process
run of the same call — assignment vectors and inlier counts identical, 4x4 matrices to 1e-9, for the replicated and the
sharded ICP.  (The N-GPU run over RCCL is the driver's; this checks the code path, not the speed.)

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
        tools/two_rank_registration.py [N [M [SEED]]]"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import synth_pair  # noqa: E402
from platymatch_amd import pipeline as P  # noqa: E402
from platymatch_amd.estimate_transform import perform_icp as pi  # noqa: E402

pi.VERBOSE = False
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
m = int(sys.argv[2]) if len(sys.argv) > 2 else n - 37       # N > M by default (assignment by gather); pass M >= N for the sharded solve
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 77
if os.environ.get("PM_SOAK_CASE") == "1":            # a case of tests/soak_cases.py (lattice clouds, duplicates, odd scales) instead of the blob
    from soak_cases import make_case
    mv, fx, _, _, _ = make_case(seed, max(n, m))
    n, m = mv.shape[1], fx.shape[1]
else:
    mv, fx, _ = synth_pair(max(n, m), seed, m=m)
    mv = np.ascontiguousarray(mv[:, :n])
if os.environ.get("PM_FILTER_FROM"):                 # the default cost mode's sharded FILTER route at rehearsal sizes (8 192 points in production)
    P.FILTER_MIN_POINTS = int(os.environ["PM_FILTER_FROM"])
kw = dict(ransac_trials=500, ransac_error=8.0, icp_iterations=12, seed=4)
streamed = os.environ.get("PM_STREAM_HYPOTHESES") == "1"       # two cost matrices resident at a time on every rank (config 4's mode)
ok = True
for shard_icp in (False, True):
    det = {}
    got = P.estimate_transform(mv, fx, group=dist.group.WORLD, details=det, options={"icp_shard_min_points": 0 if shard_icp else 10 ** 9,
                                                              "stream_hypotheses": True if streamed else None}, **kw)
    if rank == 0:
        ref_det = {}
        ref = P.estimate_transform(mv, fx, details=ref_det, cost_mode='exact', **kw)      # one process, exact matrices: the yardstick
        same_lsa = all(np.array_equal(a[1], b[1]) for a, b in zip(det["lsa"], ref_det["lsa"]))
        err_sc = np.linalg.norm(got[0] - ref[0]) / np.linalg.norm(ref[0])
        err_f = np.linalg.norm(got[1] @ got[0] - ref[1] @ ref[0]) / np.linalg.norm(ref[1] @ ref[0])
        res = np.abs(det["residuals"] - ref_det["residuals"]).max()
        good = same_lsa and np.array_equal(got[2], ref[2]) and err_sc == 0.0 and err_f < 1e-9 and res < 1e-9
        ok &= good
        print("cost mode:", det.get("assignment", {}).get("cost_mode"), "| assignment routes:", det.get("assignment", {}).get("routes"), flush=True)
        print("ICP %s: assignments identical %s, inliers identical %s, A_sc identical %s, final rel. diff %.1e, residual diff %.1e -> %s"
              % ("sharded" if shard_icp else "replicated", same_lsa, np.array_equal(got[2], ref[2]), err_sc == 0.0, err_f, res,
                 "OK" if good else "MISMATCH"), flush=True)
    dist.barrier()
flag = torch.tensor([1 if ok else 0])
dist.broadcast(flag, 0)
dist.destroy_process_group()
sys.exit(0 if int(flag) == 1 else 1)
