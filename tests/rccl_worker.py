"""Run by tests/test_gpu_rccl.py in a FRESH process: initialise torch.distributed's "nccl" backend (= RCCL on ROCm) at world
size 1 on cuda:0 and push one tensor of every (collective, dtype, reduce-op) combination the sharded pipeline uses through
it; then run the headless driver with group=WORLD on a reference fixture.  Prints one JSON line.

Collectives of the product (platymatch_amd/pipeline.py, lsap_sharded.py, bench.py):
  all_reduce SUM f64   cloud_statistics (mean-distance tile sums), estimate_transform_batch (result table)
  all_reduce MAX i32   gather_fixed_descriptors (symmetry flag), assign (status words)
  all_reduce MAX f64   bench.py (max-over-ranks step time)
  all_gather_into_tensor f64 / i32   all_gather_rows (fixed descriptors; cost_row_argmins), icp_sharded (26 doubles per iteration)
  gather f64 / i32     assign (row blocks of an uncertified hypothesis to its owner); lsap_sharded (candidates, counters, status)
  broadcast i64 / i32 / f64          assign (index vectors from the owner); lsap_sharded (header, duals, result)
  barrier              bench.py
No Python object crosses the wire (round 4): the sharded assignment's query protocol is tensor collectives only, and runs here
through RCCL with the real DeviceMatrix (solve_pair_sharded on a world of one).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29531")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    out = {"backend": dist.get_backend(), "world": dist.get_world_size()}
    g = dist.group.WORLD

    x = torch.arange(1000, dtype=torch.float64, device=dev) / 7.0
    want = x.clone()
    dist.all_reduce(x, op=dist.ReduceOp.SUM, group=g)
    out["all_reduce_sum_f64"] = bool(torch.equal(x, want))
    f = torch.tensor([3], dtype=torch.int32, device=dev)
    dist.all_reduce(f, op=dist.ReduceOp.MAX, group=g)
    out["all_reduce_max_i32"] = int(f.item()) == 3
    t = torch.tensor([1.25], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=g)
    out["all_reduce_max_f64"] = float(t.item()) == 1.25
    blk = torch.rand((1, 37, 360), dtype=torch.float64, device=dev)
    got = [torch.empty_like(blk)]
    dist.all_gather(got, blk, group=g)
    out["all_gather_f64"] = bool(torch.equal(got[0], blk))
    idx = torch.arange(8 * 11, dtype=torch.int32, device=dev).reshape(8, 11)
    goti = [torch.empty_like(idx)]
    dist.all_gather(goti, idx, group=g)
    out["all_gather_i32"] = bool(torch.equal(goti[0], idx))
    rows = torch.rand((5, 64), dtype=torch.float64, device=dev)
    whole = torch.empty((1, 5, 64), dtype=torch.float64, device=dev)
    dist.gather(rows, [whole[0]], dst=0, group=g)
    out["gather_f64"] = bool(torch.equal(whole[0], rows))
    b = torch.arange(2 * 9, dtype=torch.int64, device=dev).reshape(2, 9)
    keep = b.clone()
    dist.broadcast(b, src=0, group=g)
    out["broadcast_i64"] = bool(torch.equal(b, keep))
    into = torch.empty((37, 360), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(into, blk[0], group=g)
    out["all_gather_into_tensor_f64"] = bool(torch.equal(into, blk[0]))
    into_i = torch.empty(11, dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(into_i, idx[3], group=g)
    out["all_gather_into_tensor_i32"] = bool(torch.equal(into_i, idx[3]))
    dist.barrier(group=g)
    torch.cuda.synchronize()

    # the product's own helpers over the same group
    from platymatch_amd import pipeline as P
    from platymatch_amd.estimate_transform import perform_icp as pi
    pi.VERBOSE = False
    be = P.GpuBackend(dev)
    local = torch.rand((4, 21, 360), dtype=torch.float64, device=dev)
    out["all_gather_rows"] = bool(torch.equal(P.all_gather_rows(local, [0, 21], 1, g), local))

    # the sharded statistics step by step (pipeline.cloud_statistics takes the one-GPU shortcut at world size 1): tile sums ->
    # RCCL all-reduce -> ordered finish must give the bits of the one-device kernel; the symmetry flag through MAX
    cloud = be.cloud(np.load(os.path.join(ROOT, "tests", "golden", "synth1000.npz"))["moving"])
    part = be.mean_distance_partials(cloud, 0, 1)
    dist.all_reduce(part, op=dist.ReduceOp.SUM, group=g)
    md = be.mean_distance_finish(part, cloud.shape[1])
    out["sharded_mean_distance_bits"] = bool(torch.equal(md.reshape(-1), be.stats(cloud)[1].reshape(-1)))
    c, md1, x0 = be.stats(cloud)
    sc_m = be.shape_context(cloud, c, md1, x0, 2, 0, cloud.shape[1])
    sc_f = be.shape_context(cloud, c, md1, x0, 4, 0, cloud.shape[1])
    flag = be.symmetry_flag(sc_m, sc_f)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=g)
    out["symmetry_flag_dtype"] = str(flag.dtype)
    out["symmetry_flag"] = int(flag.item())

    # the sharded assignment's query protocol (tensor collectives only) through RCCL with the real kernels behind it: a hypothesis
    # and its twin of a 1 400-point pair by solve_pair_sharded on this world of one == SciPy on the same matrices
    from scipy.optimize import linear_sum_assignment as scipy_lsa
    from platymatch_amd import lsap as L
    from platymatch_amd.lsap_sharded import solve_pair_sharded
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import synth_pair
    mv, fx2, _ = synth_pair(1400, 9)
    U8, bn = P.build_costs(be, be.cloud(mv), be.cloud(fx2))
    sinfo = {}
    c_h, c_t = solve_pair_sharded(L.DeviceMatrix(U8[0]), L.DeviceMatrix(U8[5]), bn, U8.shape[2], g, 0, sinfo)
    out["sharded_query_protocol"] = bool(c_h is not None and c_t is not None
                                         and np.array_equal(c_h, scipy_lsa(U8[0].cpu().numpy())[1])
                                         and np.array_equal(c_t, scipy_lsa(U8[5].cpu().numpy())[1]))

    # round 5: the default mode's sharded FILTER route through RCCL (world of one: the whole row range is this rank's block) — float32
    # dense passes of DeviceMatrix answering a FilteredMatrix over the ShardedMatrix, exact entries on the root: the exact matrices'
    # eight assignments; and bench.py's self-proving record (uint8 all-gather of the device identity, the threaded preflight)
    sc_m, sc_f, bn2 = P.build_descriptors(be, be.cloud(mv), be.cloud(fx2))
    finfo = {}
    lsa_f = P.assign_sharded_filtered(be, sc_m, sc_f[:1].contiguous(), bn2, g, info=finfo)
    out["sharded_filter_route"] = bool(all(np.array_equal(lsa_f[h][1], scipy_lsa(U8[h].cpu().numpy())[1]) for h in range(8))
                                       and all("(filter" in str(r) for r in finfo["routes"]))
    lsa_s = P.assign_sharded_filtered(be, sc_m, sc_f[:1].contiguous(), bn2, g, streamed=True)
    out["sharded_filter_route_streamed"] = bool(all(np.array_equal(lsa_s[h][1], lsa_f[h][1]) for h in range(8)))
    import bench
    rec = bench.rank_devices(dist, g, dev, 1, 0)
    out["rank_devices"] = bool(rec["backend"] == "nccl" and rec["distinct_devices"] == 1 and len(rec["devices"][0]) > 3)
    out["preflight_s"] = float(bench.preflight(dist, g, dev, 1, timeout_s=120.0))

    # the headless driver with group=WORLD against the reference fixture (tests/golden/synth128.npz)
    fx = np.load(os.path.join(ROOT, "tests", "golden", "synth128.npz"))
    det = {}
    A_sc, A_icp, inl = P.estimate_transform(fx["moving"], fx["fixed"], ransac_trials=int(fx["ransac_trials"]),
                                            ransac_error=float(fx["ransac_error"]), icp_iterations=int(fx["icp_iters"]),
                                            seed=int(fx["ransac_seed"]), details=det, group=g)
    out["lsa_equal_fixture"] = all(bool(np.array_equal(det["lsa"][h][1], fx["lsa_cols"][h]) and np.array_equal(det["lsa"][h][0], fx["lsa_rows"][h]))
                                   for h in range(8))
    out["inliers_equal_fixture"] = bool(np.array_equal(inl, fx["ransac_inliers"]))
    ref = fx["A_final"]
    out["A_final_relerr"] = float(np.linalg.norm(A_icp @ A_sc - ref) / np.linalg.norm(ref))
    dist.barrier(group=g)
    dist.destroy_process_group()
    print("RCCL_WORKER " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
