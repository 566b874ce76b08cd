#!/usr/bin/env python3
"""The gate of cost_mode='auto' as the DEFAULT (VERDICT r04 next #1c): the eight assignment vectors of the default mode against
cost_mode='exact' on a stream of adversarial registrations — every case must give identical (row_ind, col_ind) for all eight
hypotheses (or raise the same exception in both modes).  Families:
  generic          anisotropic blobs, N = M and N != M both ways, jitter 0 / 0.3 / 1, scales 1e-3 .. 1e3
  near-duplicates  3..60 nuclei of each cloud replaced by copies of others displaced by 1e-12 .. 1e-4 of the cloud's size: cost rows
                   that differ from the 4th to the 15th digit — alternatives far inside, at and above the certificate's margins
  lattice          half-integer lattice coordinates (neighbours exactly on ring / sector edges, tied distances, duplicates)
  same cloud twice the fixed cloud is a permutation of the moving one (zero-cost matches, exact ties)
  planar           all z equal (degenerate frames, many boundary hits)
  mirrored         a cloud symmetric under a reflection (pairs of nuclei with bit-identical descriptors: exact ties by construction)
Tie-prone families stay below --tie-max points, near-duplicates below 6 000 (the exact mode settles ties with the dense host
solver: minutes at 20 000); generic clouds go up to --max-points.
Worker threads drive independent cases on their own HIP streams.
Usage: python tools/auto_soak.py --seconds 600 [--cases N] [--max-points 20000] [--tie-max 2500] [--seed0 0] [--workers 4]
                                 [--filter-from 8192]      (lower it to push small cases through the filter route too)
                                 [--min-points 30000 --max-points 60000 --workers 1]      (large cases: the exact mode's 64 N M bytes per case)"""
import argparse
import os
import sys
import threading
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from platymatch_amd import _native as nat, pipeline as P  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=300.0)
ap.add_argument("--cases", type=int, default=10 ** 9)
ap.add_argument("--max-points", type=int, default=20000)
ap.add_argument("--min-points", type=int, default=0)      # > 0: LARGE cases only — generic clouds and near-duplicates displaced by >= 1e-7, sizes uniform in [min, max]
ap.add_argument("--tie-max", type=int, default=2500)
ap.add_argument("--seed0", type=int, default=0)
ap.add_argument("--workers", type=int, default=4)
ap.add_argument("--filter-from", type=int, default=P.FILTER_MIN_POINTS)
args = ap.parse_args()
P.FILTER_MIN_POINTS = max(P.RELAXED_MIN_POINTS, args.filter_from)
LO = P.RELAXED_MIN_POINTS
FAMILIES = ("generic", "near-duplicates", "lattice", "same cloud twice", "planar", "mirrored")


def make(seed):
    rng = np.random.default_rng(7919 * seed + 11)
    kind = FAMILIES[int(rng.choice(len(FAMILIES), p=[0.30, 0.34, 0.10, 0.08, 0.08, 0.10]))]
    if args.min_points:
        kind = "generic" if rng.random() < 0.6 else "near-duplicates"
    # (near-duplicates carry nearly or exactly tied cost rows in all eight hypotheses: where the exact mode cannot prove uniqueness
    # it runs the dense host solver — seconds at 6 000 points, minutes at 20 000)
    hi = args.max_points if kind == "generic" else (min(args.max_points, 6000) if kind == "near-duplicates" else min(args.max_points, args.tie_max))
    # sizes log-uniform in [LO, hi], biased to the small end (throughput); one case in 12 of the large families near the top
    def size():
        if hi > 8192 and rng.random() < 1.0 / 12.0:
            return int(rng.integers(8192, hi + 1))
        return int(np.exp(rng.uniform(np.log(LO), np.log(min(hi, 6000)))))
    if args.min_points:
        size = lambda: int(rng.integers(args.min_points, args.max_points + 1))      # noqa: E731
    n = size()
    m = n if rng.random() < 0.35 else size()
    big = max(n, m)
    axes = rng.uniform(5.0, 80.0, size=(3, 1)) * rng.choice([1.0, 1.0, 0.3], size=(3, 1))
    base = rng.normal(size=(3, big)) * axes + rng.uniform(-300.0, 300.0, size=(3, 1))
    th = rng.uniform(-0.6, 0.6, size=3)
    Rz = np.array([[np.cos(th[0]), -np.sin(th[0]), 0], [np.sin(th[0]), np.cos(th[0]), 0], [0, 0, 1]])
    Ry = np.array([[np.cos(th[1]), 0, np.sin(th[1])], [0, 1, 0], [-np.sin(th[1]), 0, np.cos(th[1])]])
    A = (Rz @ Ry) * rng.uniform(0.7, 1.4) + rng.normal(scale=0.03, size=(3, 3))
    t = rng.uniform(-50.0, 50.0, size=(3, 1))
    if kind == "planar":
        base[2] = base[2, 0]
    if kind == "mirrored":                                   # symmetric under x -> -x about the cloud's own plane
        half = big // 2
        base[:, half:2 * half] = base[:, :half] * np.array([[1.0], [1.0], [-1.0]]) + np.array([[0.0], [0.0], [2.0 * base[2].mean()]])
    mv = base.copy()
    fx = A @ base + t + rng.normal(scale=rng.choice([0.0, 0.3, 1.0]), size=(3, big))
    if kind == "planar":
        fx[2] = fx[2, 0]
    if kind == "mirrored":
        fx = A @ base + t                                    # no jitter: the symmetry (and its ties) survive the map
    if kind == "lattice":
        mv, fx = np.round(mv * 0.1) * 5.0, np.round(fx * 0.1) * 5.0
    fx = fx[:, rng.permutation(big)]
    mv, fx = np.ascontiguousarray(mv[:, :n]), np.ascontiguousarray(fx[:, :m])
    if kind == "same cloud twice":
        fx = np.ascontiguousarray(mv[:, rng.permutation(n)])
    if kind == "near-duplicates":
        k = int(rng.integers(3, 61))
        for cloud in (mv, fx):
            size_ = float(np.abs(cloud).max())
            kk = min(k, cloud.shape[1] // 3)
            src = rng.choice(cloud.shape[1], size=kk, replace=False)
            dst = rng.choice(cloud.shape[1], size=kk, replace=False)
            cloud[:, dst] = cloud[:, src] + rng.normal(size=(3, kk)) * size_ * 10.0 ** rng.uniform(-7 if args.min_points else -12, -4, size=(1, kk))
    scale = float(rng.choice([1.0, 1.0, 1e-3, 1e3]))
    return np.ascontiguousarray(mv * scale), np.ascontiguousarray(fx * scale), kind


lock = threading.Lock()
state = {"next": args.seed0, "done": 0}
counts, fails = {}, []
t_end = time.perf_counter() + args.seconds


def run_mode(mv, fx, mode, det):
    try:
        return P.assignments(mv, fx, cost_mode=mode, details=det), None
    except Exception as e:      # noqa: BLE001 — compared between the modes
        return None, "%s: %s" % (type(e).__name__, str(e)[:160])


def worker(slot):
    import torch
    dev = nat.device()
    stream = nat.side_stream(dev, ("auto soak", slot))
    with torch.cuda.device(dev), torch.cuda.stream(stream), warnings.catch_warnings():
        warnings.simplefilter("ignore", P.EdgeGuardWarning)
        while True:
            with lock:
                if time.perf_counter() > t_end or state["done"] >= args.cases:
                    return
                seed = state["next"]
                state["next"] += 1
            mv, fx, kind = make(seed)
            tag = "seed %d (%s, N=%d, M=%d)" % (seed, kind, mv.shape[1], fx.shape[1])
            de, da = {}, {}
            want, err_e = run_mode(mv, fx, "exact", de)
            got, err_a = run_mode(mv, fx, "auto", da)
            stream.synchronize()
            bad = None
            if (want is None) != (got is None):
                bad = tag + ": one mode raised (%s) and the other did not (%s)" % (err_e, err_a)
            elif want is None:
                if err_e.split(":")[0] != err_a.split(":")[0]:
                    bad = tag + ": different exceptions: %s / %s" % (err_e, err_a)
            elif not all(np.array_equal(want[h][0], got[h][0]) and np.array_equal(want[h][1], got[h][1]) for h in range(8)):
                which = [h for h in range(8) if not (np.array_equal(want[h][0], got[h][0]) and np.array_equal(want[h][1], got[h][1]))]
                bad = tag + ": assignment vectors differ for hypotheses %s" % which
            modes = [str(d.get("cost_mode", "")) for d in da.get("assignment", {}).get("details", [])]
            with lock:
                c = counts.setdefault(kind, dict(cases=0, relaxed=0, filter=0, exact=0, raised=0, largest=0))
                c["cases"] += 1
                c["relaxed"] += sum(x.startswith("relaxed") for x in modes)
                c["filter"] += sum(x.startswith("filter") for x in modes)
                settled = sum(x.startswith(("relaxed", "filter")) for x in modes)
                c["exact"] += (8 - settled) if got is not None else 0       # (incl. pairs exact from the start: small, or frames that do not permute)
                c["raised"] += int(want is None)
                c["largest"] = max(c["largest"], min(mv.shape[1], fx.shape[1]))
                state["done"] += 1
                if bad:
                    fails.append(bad)
                    print("MISMATCH " + bad, flush=True)
                if args.min_points:
                    print("    %s: %s" % (tag, "raised alike (%s)" % err_e if want is None else "; ".join(sorted(set(modes)))), flush=True)


threads = [threading.Thread(target=worker, args=(k,)) for k in range(max(1, args.workers))]
t0 = time.perf_counter()
for th in threads:
    th.start()
while any(th.is_alive() for th in threads):
    time.sleep(30.0)
    with lock:
        print("... %d cases in %.0f s, %d mismatches" % (state["done"], time.perf_counter() - t0, len(fails)), flush=True)
for th in threads:
    th.join()
print("auto soak: seeds %d..%d, %d cases in %.0f s on %d worker threads; %d..%d points (tie-prone families <= %d); filter from %d points"
      % (args.seed0, state["next"] - 1, state["done"], time.perf_counter() - t0, args.workers, LO, args.max_points, args.tie_max, P.FILTER_MIN_POINTS))
for kind, c in sorted(counts.items()):
    print("  %-18s %5d cases (smaller cloud up to %5d): hypotheses settled on the relaxed build %6d, through the filter %6d, by an exact "
          "build %6d; both modes raised alike %d" % (kind, c["cases"], c["largest"], c["relaxed"], c["filter"], c["exact"], c["raised"]))
print("assignment differences against cost_mode='exact': %d" % len(fails))
for f in fails[:60]:
    print("  " + f)
sys.exit(1 if fails else 0)
