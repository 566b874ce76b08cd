"""`python bench.py --gpus N` as the driver starts it: ONE process, no WORLD_SIZE in the environment.  The script must
become a launcher — N fresh children of itself with the rendezvous environment of torch.distributed.run — before torch is
imported or a GPU is touched, relay rank 0's JSON line and return the worst child's exit code."""
import json
import os
import subprocess
import sys

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "PM_BENCH_SPAWNED")}
    env.update(extra)
    return env


def _json_lines(text):
    return [json.loads(l) for l in text.splitlines() if l.startswith("{")]


def test_two_ranks_are_spawned_and_rendezvous_over_gloo():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run"], env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1, p.stdout                       # rank 0 only
    ranks = lines[0].pop("ranks")
    assert lines[0] == {"dry_run": True, "n_gpus": 2, "max_rank_plus_one": 2.0, "spawned": True, "local_rank": 0}
    assert ranks["backend"] == "gloo" and ranks["world"] == 2 and ranks["distinct_devices"] == 2 and len(ranks["devices"]) == 2


def test_eight_ranks_are_spawned_and_rendezvous_over_gloo():
    """The driver's scaling run at its widest (--gpus 8): eight children rendezvous (gloo, dry run, no GPU), rank 0 alone reports."""
    p = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--dry-run"], env=_clean_env(OMP_NUM_THREADS="1"), capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1, p.stdout
    ranks = lines[0].pop("ranks")
    assert lines[0] == {"dry_run": True, "n_gpus": 8, "max_rank_plus_one": 8.0, "spawned": True, "local_rank": 0}
    # the self-proving record of a multi-rank line: which backend carried the collectives, how many DISTINCT
    # devices the ranks drive (all-gathered identities), and that every collective of the step passed the 60 s preflight
    assert ranks["backend"] == "gloo" and ranks["world"] == 8 and ranks["distinct_devices"] == 8
    assert ranks["devices"] == ["no device (dry run), rank %d" % r for r in range(8)] and 0.0 < ranks["preflight_s"] < 60.0


def test_preflight_turns_a_missing_rank_into_an_error_not_a_hang():
    """One of two ranks never arrives: the other's collective preflight gives up after its timeout with a message naming the
    rendezvous variables (here 3 s; 60 s in a real run) instead of waiting inside the timed region."""
    code = ("import os, sys, time\nsys.path.insert(0, %r)\nimport bench\nimport torch.distributed as dist\n"
            "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=sys.argv[2])\n"
            "rank = int(sys.argv[1])\ndist.init_process_group('gloo', rank=rank, world_size=2)\n"
            "if rank == 1:\n    time.sleep(20)\n    os._exit(0)\n"
            "bench.preflight(dist, dist.group.WORLD, None, 2, timeout_s=3.0)\n" % ROOT)
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    late = subprocess.Popen([sys.executable, "-c", code, "1", port], env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    p = subprocess.run([sys.executable, "-c", code, "0", port], env=_clean_env(), capture_output=True, text=True, timeout=120)
    late.kill()
    late.wait()
    assert p.returncode == 3 and "collective preflight did not finish" in p.stderr and "MASTER_ADDR" in p.stderr


def test_launcher_does_not_import_torch():
    """The parent must stay clear of the GPU runtime: it may not even import torch."""
    code = ("import sys, runpy; sys.argv = ['bench.py', '--gpus', '2', '--dry-run']\n"
            "try:\n    runpy.run_path(%r, run_name='__main__')\nexcept SystemExit as e:\n    rc = e.code\n"
            "assert 'torch' not in sys.modules, 'launcher imported torch'\nsys.exit(rc)\n" % BENCH)
    p = subprocess.run([sys.executable, "-c", code], env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert _json_lines(p.stdout)[0]["n_gpus"] == 2


def test_a_failing_rank_fails_the_launcher_without_a_gpu():
    """Without --dry-run on a box with fewer devices than ranks every missing device is reported by its rank ("device k not
    found") and the launcher's exit code is non-zero — it does not stop at a WORLD_SIZE check."""
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("two devices present")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--points", "512"],
                       env=_clean_env(PM_BENCH_SPAWN_GRACE_S="5"), capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    assert "device 1 not found" in p.stderr
    assert "launch with torch.distributed.run" not in p.stderr


def test_mismatched_world_size_is_refused():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run"], env=_clean_env(WORLD_SIZE="3", RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=3" in p.stderr
