"""The sharded path with the REAL GPU backend on several ranks: the ranks share this box's one GPU and gloo carries the
collectives (two ranks cannot share a device under RCCL; RCCL itself runs at world size 1 in tests/test_gpu_rccl.py, and the
N-GPU run is the driver's).  What is checked is the code every rank executes — row-sharded statistics, descriptors, cost blocks,
the sharded assignment's query protocol over DeviceMatrix blocks, replicated and sharded ICP, the streamed mode — against the
one-process run of the same call: identical assignment vectors, inlier counts and A_sc, final 4x4 to 1e-9
(tools/two_rank_registration.py); and bench.py's own multi-rank step."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(extra)
    return env


def _ranks(world, args, **extra):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tools", "two_rank_registration.py")] + [str(a) for a in args]
    return subprocess.run(cmd, env=_env(**extra), capture_output=True, text=True, timeout=900, cwd=ROOT)


@pytest.mark.parametrize("world,n,m,streamed", [(2, 1500, 1600, False), (2, 1600, 1500, False), (3, 1300, 1400, True), (3, 1400, 1300, True)])
def test_ranks_sharing_the_gpu_reproduce_the_one_process_registration(world, n, m, streamed):
    """N <= M: the sharded device solve (nothing gathered); N > M: gathered to the hypothesis's owner, or — streamed — the
    transposed roles of round 3.  Both ICP variants (replicated, sharded) inside the script."""
    p = _ranks(world, [n, m], **({"PM_STREAM_HYPOTHESES": "1"} if streamed else {}))
    assert p.returncode == 0, (p.stdout[-2500:], p.stderr[-2500:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("ICP ")]
    assert len(lines) == 2 and all(l.endswith("OK") for l in lines), p.stdout[-2500:]


@pytest.mark.parametrize("world,n,m,streamed", [(2, 2500, 2700, False), (3, 2700, 2500, False), (3, 2400, 2400, True)])
def test_sharded_filter_route_on_ranks_sharing_the_gpu(world, n, m, streamed):
    """The DEFAULT cost mode on several ranks (VERDICT r04 next #1b): every rank builds its row block of the float32 filter
    (pm_chi2_filter4_f32 on its rows; N > M: its block of FIXED rows, roles swapped), the root's solver is answered from the blocks
    (float32 dense passes of lsap.DeviceMatrix) and evaluates exact entries itself (pm_chi2_entries_sym on the gathered frame-1
    descriptors).  Threshold lowered to rehearsal size; yardstick: the one-process registration on the EXACT matrices."""
    p = _ranks(world, [n, m], PM_FILTER_FROM="1024", **({"PM_STREAM_HYPOTHESES": "1"} if streamed else {}))
    assert p.returncode == 0, (p.stdout[-2500:], p.stderr[-2500:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("ICP ")]
    assert len(lines) == 2 and all(l.endswith("OK") for l in lines), p.stdout[-2500:]
    assert "cost mode: filter" in p.stdout and "sharded device (filter" in p.stdout and "'gathered'" not in p.stdout, p.stdout[-2500:]


def test_sharded_filter_route_on_a_lattice_cloud_builds_exactly_what_it_cannot_prove():
    """Case 29 again (exact ties everywhere) through the default mode's sharded filter route: no pairing can be certified on listed
    entries, each gets its exact row blocks and goes the exact sharded way — the answers stay the one-process exact run's."""
    p = _ranks(3, [2600, 2600, 29], PM_SOAK_CASE="1", PM_FILTER_FROM="1024")
    assert p.returncode == 0, (p.stdout[-2500:], p.stderr[-2500:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("ICP ")]
    assert len(lines) == 2 and all(l.endswith("OK") for l in lines), p.stdout[-2500:]
    assert "cost mode: filter" in p.stdout or "cost mode: exact" in p.stdout      # (exact from the start where the frames do not permute)


def test_sharded_run_on_a_lattice_cloud_gathers_what_ties_leave_uncertified():
    """tests/soak_cases.py case 29: a coarse lattice with duplicates — exact ties in every cost matrix.  Streamed over three
    ranks the sharded solve cannot certify a unique optimum; the pairing's blocks then go to the hypothesis's owner and SciPy's
    algorithm answers (round 3; before: a refusal).  Same assignments, inlier counts and A_sc as the one-process run."""
    p = _ranks(3, [2600, 2600, 29], PM_SOAK_CASE="1", PM_STREAM_HYPOTHESES="1")
    assert p.returncode == 0, (p.stdout[-2500:], p.stderr[-2500:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("ICP ")]
    assert len(lines) == 2 and all(l.endswith("OK") for l in lines), p.stdout[-2500:]
    assert "'gathered'" in p.stdout


def test_sharded_icp_on_a_nearly_planar_cloud_falls_back_to_the_replicated_loop():
    """tests/soak_cases.py case 3: one axis 0.2 of the others and a transform that flattens it further — the sharded ICP's normal
    equations are too ill-conditioned (pm_solve.h: PM_DEGENERATE_MOMENTS); every rank leaves it together and runs the replicated
    loop with the reference's pinv fits (round 3; before: a refusal).  Equal to the one-process run."""
    p = _ranks(2, [2600, 2600, 3], PM_SOAK_CASE="1")
    assert p.returncode == 0, (p.stdout[-2500:], p.stderr[-2500:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("ICP ")]
    assert len(lines) == 2 and all(l.endswith("OK") for l in lines), p.stdout[-2500:]


def test_bench_step_on_two_ranks_sharing_the_gpu():
    """`python bench.py --gpus 2` through its own launcher, the ranks put on cuda:0 and gloo in RCCL's place (rehearsal
    switches of bench.py): the sharded step end to end, one JSON line from rank 0."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--points", "6000",
                        "--icp-iters", "5", "--no-cpu-baseline"], env=_env(PM_BENCH_ONE_DEVICE="1", PM_BENCH_BACKEND="gloo"),
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-2500:])
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1
    d = json.loads(line[0])
    assert d["n_gpus"] == 2 and d["config"]["sharding"] == "rows/2" and d["value"] > 0 and d["scaling"] == "strong"
    assert set(d["stage_ms"]) == {"statistics", "shape_context", "chi2_cost8", "icp"}
    # the line proves what ran: backend, world, and that BOTH ranks sat on one device — flagged as the rehearsal
    # it is (without the switch bench.py refuses repeated devices)
    r = d["ranks"]
    assert r["backend"] == "gloo" and r["world"] == 2 and r["distinct_devices"] == 1 and r["rehearsal_on_one_device"] is True
    assert len(r["devices"]) == 2 and r["devices"][0] == r["devices"][1] and 0.0 < r["preflight_s"] < 60.0
    # a sharded line carries its roofline (rank 0's row block; counter traffic explicitly null with the reason) ...
    rf = d["roofline"]
    assert rf["rows_of_this_rank"] == 3000 and rf["algorithmic_bytes"] > 0 and rf["traffic"] is None and rf["traffic_why_null"]
    # ... and the eight assignments by the sharded default route (row blocks of the float32 filter), timed over both ranks
    a = d["assignment_extra"]
    assert a["error"] is None and a["route"].startswith("sharded filter") and a["seconds"] > 0 and all(a["perfect_matchings"])
    assert all("(filter" in str(x) for x in a["routes"]), a["routes"]


def test_bench_line_survives_a_sharded_extra_that_does_not_return():
    """The sharded assignment extra runs under a watchdog (bench.py): with the limit set to nothing it "hangs" by definition — rank 0
    must still print its one JSON line (the extra reported as timed out, no CPU baseline touching the device afterwards) and every
    rank must leave with exit code 0 without the closing barrier."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--points", "6000",
                        "--icp-iters", "3"], env=_env(PM_BENCH_ONE_DEVICE="1", PM_BENCH_BACKEND="gloo", PM_BENCH_EXTRA_TIMEOUT_S="0.001"),
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-2500:])
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1
    d = json.loads(line[0])
    assert d["value"] > 0 and d["n_gpus"] == 2
    # rank 0 either sees its own watchdog fire ("timed out", no CPU baseline afterwards) or — when rank 1's fired first and rank 1 has
    # left already — a collective of its extra fail at once ("Connection closed by peer"): the extra is lost either way, the line is not
    err = d["assignment_extra"]["error"]
    assert err and d["assignment_extra"]["seconds"] is None
    assert ("timed out" in err and "skipped" in d["cpu_baseline"]) or "value" in d["cpu_baseline"], (err, d["cpu_baseline"])
