"""The registration driver's mode matrix, one cell per test id so that a missing cell is visible:

    cost mode   exact | relaxed | filter          (what 'auto' resolves to is asserted separately, size by size)
    residency   resident (all matrices of the build in HBM) | streamed (pairing by pairing)
    sharding    one GPU | two ranks (sharing this box's GPU, gloo carrying the collectives: tools/two_rank_registration.py)
    sampler     seeded (the reference's MT19937 index sets) | unseeded (device sampler)     [one-GPU cells]

Every cell must return the yardstick's eight assignment vectors; seeded cells also its inlier counts and 4 x 4 matrices bit for
bit (the yardstick: cost_mode='exact', resident, one GPU, seeded); unseeded cells recover the ground-truth transform.  Cells that
do not exist as separate code — 'relaxed' has no sharded form — are asserted to resolve to the cell that runs instead."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, synth_pair

pytestmark = pytest.mark.gpu
N, M = 1800, 1900
KW = dict(ransac_trials=300, ransac_error=16, icp_iterations=6)


@pytest.fixture(scope="module")
def yardstick():
    import torch
    from platymatch_amd import _native as nat, pipeline as P
    from platymatch_amd.estimate_transform import perform_icp as pi
    nat.load()
    assert torch.cuda.is_available()
    pi.VERBOSE = False
    mv, fx, A_gt = synth_pair(M, 71)
    mv = np.ascontiguousarray(mv[:, :N])
    det = {}
    out = P.estimate_transform(mv, fx, seed=11, details=det, cost_mode='exact', options={"stream_hypotheses": False}, **KW)
    return dict(P=P, mv=mv, fx=fx, A_gt=A_gt, out=out, lsa=det["lsa"])


@pytest.mark.parametrize("seeded", [True, False], ids=["seeded", "unseeded"])
@pytest.mark.parametrize("streamed", [False, True], ids=["resident", "streamed"])
@pytest.mark.parametrize("mode", ["exact", "relaxed", "filter"])
def test_one_gpu_cell(yardstick, mode, streamed, seeded, monkeypatch):
    y = yardstick
    P = y["P"]
    if mode == "filter":
        monkeypatch.setattr(P, "FILTER_MIN_POINTS", P.RELAXED_MIN_POINTS)       # (8 192 as shipped: the cell is exercised at test size)
    det = {}
    out = P.estimate_transform(y["mv"], y["fx"], seed=11 if seeded else None, details=det, cost_mode=mode,
                               options={"stream_hypotheses": streamed}, **KW)
    a = det["assignment"]
    # the cell that ran is the cell that was asked for ('relaxed' streamed is the exact build pairing by pairing: there is no
    # streamed relaxed build — two relaxed matrices at a time would save nothing over two exact ones)
    assert a["cost_mode"] == mode
    assert ("streamed" in str(a.get("mode", ""))) == streamed, a.get("mode")
    for h in range(8):
        assert np.array_equal(det["lsa"][h][0], y["lsa"][h][0]) and np.array_equal(det["lsa"][h][1], y["lsa"][h][1]), h
    if seeded:
        assert np.array_equal(out[2], y["out"][2]) and np.array_equal(out[0], y["out"][0]) and np.array_equal(out[1], y["out"][1])
    else:
        assert np.linalg.norm(out[1] @ out[0] - y["A_gt"]) / np.linalg.norm(y["A_gt"]) < 5e-3 and out[2].max() > 0.9 * N


def test_what_auto_resolves_to_size_by_size(yardstick):
    P = yardstick["P"]
    be = P.GpuBackend
    assert [P.resolve_cost_mode('auto', n, n, 1, be) for n in (200, 1023, 1024, 8191, 8192, 50000)] == \
        ['exact', 'exact', 'relaxed', 'relaxed', 'filter', 'filter']
    assert P.resolve_cost_mode('auto', 50000, 900, 1, be) == 'exact'                  # the smaller cloud decides
    # sharded: the filter route from 8 192 nuclei, the exact sharded route below ('relaxed' has no sharded form)
    assert [P.resolve_cost_mode('auto', n, n, 8, be) for n in (1024, 8191, 8192, 200000)] == ['exact', 'exact', 'filter', 'filter']
    assert [P.resolve_cost_mode('relaxed', n, n, 8, be) for n in (1024, 50000)] == ['exact', 'exact']
    assert P.resolve_cost_mode('exact', 50000, 50000, 1, be) == 'exact' and P.resolve_cost_mode('filter', 50000, 50000, 1, be) == 'filter'
    with pytest.raises(ValueError):
        P.resolve_cost_mode('fast', 10, 10, 1, be)


def _two_ranks(args, **extra):
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tools", "two_rank_registration.py")] + [str(a) for a in args]
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)


@pytest.mark.parametrize("streamed", [False, True], ids=["resident", "streamed"])
@pytest.mark.parametrize("mode", ["exact", "filter"])
def test_two_rank_cell(mode, streamed):
    """Sharded cells (the yardstick inside the script: the one-process registration on the exact matrices; replicated and sharded
    ICP both).  'exact' = the default below 8 192 nuclei; 'filter' = the default from 8 192 on, threshold lowered to test size."""
    extra = {"PM_FILTER_FROM": "1024"} if mode == "filter" else {}
    if streamed:
        extra["PM_STREAM_HYPOTHESES"] = "1"
    p = _two_ranks([N, M], **extra)
    assert p.returncode == 0, (p.stdout[-2500:], p.stderr[-2500:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("ICP ")]
    assert len(lines) == 2 and all(l.endswith("OK") for l in lines), p.stdout[-2500:]
    assert ("cost mode: %s" % mode) in p.stdout, p.stdout[-2500:]
    assert ("sharded device (filter" in p.stdout) == (mode == "filter"), p.stdout[-2500:]
