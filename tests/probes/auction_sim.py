#!/usr/bin/env python3
"""What would an eps-scaling auction ON THE DEVICE buy the assignment stage?  (a probe, CPU only — it lives
under tests/ because the cost matrices come from the oracle.)

The device can only run the auction in its JACOBI form (every unassigned row bids at once, one round = one grid-wide step); the
host core (csrc/pm_lsap_core.cpp) runs the Gauss-Seidel form, one bid after the other at ~30 ns each.  This script builds real
chi-square matrices (BASELINE's synthetic recipe, hypotheses 11 and 12 = a right-frame and a wrong-frame one), takes the sparse
core the product would take (column reduction, 16 entries per row by reduced cost + the diagonal) and
  1. runs the Jacobi auction with the product's eps schedule and reports, per eps phase, rounds, bids and how many rounds had
     more than 1 024 / 256 / 64 bidders — the part a GPU can do in parallel — against the narrow rounds (eviction chains and
     price wars of a handful of rows), which are sequential by nature: a device round costs >= 1-3 us whatever its width;
  2. drives the REAL host solver from the device-like result: wide rounds only (a phase ends when <= 0.5 % of the rows are
     unassigned), pricing + appended entries between the passes as the product does, then pm_lsap_core_auction_resume for a
     bounded tail and the shortest-path solve + pricing rounds as in lsap.solve_core — and reports the bids and Dijkstra steps
     the host is left with, next to the all-host flow's.
Usage: python tests/probes/auction_sim.py [N ...]      (20 000 needs ~10 GB and ~15 minutes: the dense passes are NumPy here)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from conftest import synth_pair  # noqa: E402
from test_lsap_core import HostMatrix  # noqa: E402
from platymatch_amd import lsap as L  # noqa: E402


def cost_matrices(n):
    oracle.build()
    oracle.set_threads(os.cpu_count() or 1)
    mv, fx, _ = synth_pair(n, 42)
    cm, mdm, x0m = oracle.get_centroid(mv, False), oracle.get_mean_distance(mv, False), oracle.pca_axis(mv.T)
    cf, mdf, x0f = oracle.get_centroid(fx, False), oracle.get_mean_distance(fx, False), oracle.pca_axis(fx.T)
    om = oracle.normalise_counts(*oracle.shape_context_counts(cm, mdm, mv, "moving", x0=x0m))
    of = oracle.normalise_counts(*oracle.shape_context_counts(cf, mdf, fx, "fixed", x0=x0f))
    return [np.asarray(oracle.unary_distance_matrix(om[0], of[b])) for b in (0, 1)]


def jacobi(ecol, ecost, price, eps_list, cut_free, stats, log=None):
    """Synchronous auction rounds over the core (ecol / ecost [n, K], -1 = no edge) until at most cut_free rows are unassigned."""
    n, nc = ecol.shape[0], price.size
    owner = np.full(nc, -1)
    valid = ecol >= 0
    ec = np.where(valid, ecol, 0)
    assigned = None
    for eps in eps_list:
        assigned = np.full(n, -1)
        owner[:] = -1
        hist = []
        while True:
            free = np.flatnonzero(assigned < 0)
            if free.size <= cut_free:
                break
            val = np.where(valid[free], ecost[free] + price[ec[free]], np.inf)
            o = np.argsort(val, axis=1)[:, :2]
            ar = np.arange(free.size)
            b1, b2 = val[ar, o[:, 0]], val[ar, o[:, 1]]
            bj = ec[free, o[:, 0]]
            newp = price[bj] + (b2 - b1) + eps
            order = np.lexsort((free, -newp, bj))                      # per column: highest bid, lowest row on ties
            bjs = bj[order]
            first = np.ones(order.size, bool)
            first[1:] = bjs[1:] != bjs[:-1]
            w = order[first]
            wr, wc = free[w], bj[w]
            prev = owner[wc]
            assigned[prev[prev >= 0]] = -1
            owner[wc], assigned[wr], price[wc] = wr, wc, newp[w]
            hist.append(free.size)
        hs = np.array(hist, dtype=np.int64)
        stats["rounds"] += len(hist)
        stats["bids"] += int(hs.sum())
        stats["wide_bids"] += int(hs[hs > 64].sum())
        if log is not None:
            log.append("    eps %.1e x width: %6d rounds, %8d bids; rounds with > 1024 / 256 / 64 bidders: %d / %d / %d; bids in rounds of <= 64: %d"
                       % (eps / stats["width"], len(hist), hs.sum(), (hs > 1024).sum(), (hs > 256).sum(), (hs > 64).sum(), hs[hs <= 64].sum()))
    return price, assigned


def core_of(M, K=16, KMAX=40):
    nr, nc = M.shape
    v0 = M.col_min()
    cols, costs, _ = M.row_select(v0, K)
    safety = M.diagonal(nr)
    ecol = np.full((nr, KMAX), -1, np.int64)
    ecost = np.full((nr, KMAX), np.inf)
    ecol[:, :K], ecost[:, :K] = cols, costs
    has = (cols == np.arange(nr)[:, None]).any(1)
    ecol[~has, K], ecost[~has, K] = np.arange(nr)[~has], safety[~has]
    cnt = np.where(has, K, K + 1)
    scale = max(float(np.abs(costs[cols >= 0]).max()), float(np.abs(safety).max()), float(np.abs(v0).max()), 1e-300)
    spread = (costs[:, K - 1] - v0[cols[:, K - 1]]) - (costs[:, 0] - v0[cols[:, 0]])
    return v0, ecol, ecost, cnt, L.REL_DELTA * scale, float(np.mean(spread))


def eps_schedule(e0, width):
    A, out, e = L.AUCTION, [], e0
    while True:
        out.append(e)
        if e <= A["eps_min"] * width:
            return out
        e = max(e / A["factor"], A["eps_min"] * width)


def device_like_flow(M, cut_frac, resume_bids_per_row, kp=8):
    """-> info of the flow 'wide Jacobi rounds (device) + bounded tail and exact finish (host)'."""
    A = L.AUCTION
    nr, nc = M.shape
    v0, ecol, ecost, cnt, delta, width = core_of(M)
    cut = max(16, int(cut_frac * nr))
    stats = dict(bids=0, rounds=0, wide_bids=0, width=width)
    info = {}
    price = -v0.copy()
    price, assigned = jacobi(ecol, ecost, price, eps_schedule(A["eps0"] * width, width), cut, stats)
    for a_round in range(A["rounds"] - 1):
        pc, pcost, _ = M.row_select(-price, kp)
        valid = ecol >= 0
        u = np.where(valid, ecost + price[np.where(valid, ecol, 0)], np.inf).min(1)
        off = (pc >= 0) & ((pcost + price[np.maximum(pc, 0)]) < (u - delta)[:, None])
        viol = int(off.any(1).sum())
        info.setdefault("auction_violated", []).append(viol)
        for i in np.flatnonzero(off.any(1)):
            for t in np.flatnonzero(off[i]):
                if pc[i, t] not in ecol[i, :cnt[i]] and cnt[i] < ecol.shape[1]:
                    ecol[i, cnt[i]], ecost[i, cnt[i]] = pc[i, t], pcost[i, t]
                    cnt[i] += 1
        if viol <= A["stop_below"] * nr:
            break
        price, assigned = jacobi(ecol, ecost, price, eps_schedule(A["later_eps0"] * width, width), cut, stats)
    info.update(device_rounds=stats["rounds"], device_bids=stats["bids"], unassigned_after_device=int((assigned < 0).sum()))
    with L._Core(nr, nc) as core:
        core.add(ecol.astype(np.int32), np.where(ecol >= 0, ecost, 0.0))
        t0 = time.perf_counter()
        info["host_tail_bids"] = core.auction_resume(price, assigned, A["eps_min"] * width, A["eps_min"] * width, A["factor"], int(resume_bids_per_row * nr))
        info["host_tail_seconds"] = time.perf_counter() - t0
        t_core = 0.0
        while True:
            ts = time.perf_counter()
            core.solve()
            t_core += time.perf_counter() - ts
            u, v, c4r, st = core.get()
            pc, pcost, _ = M.row_select(v, kp)
            violated = core.reprice(pc, pcost, delta)
            info.setdefault("violated_per_round", []).append(violated)
            if violated == 0:
                break
        info.update(host_steps=st[1], host_augmentations=st[2], host_solve_seconds=t_core)
    info["certified_unique"] = bool(L.certify(M, u, v, c4r))
    return info


if __name__ == "__main__":
    for n in [int(a) for a in sys.argv[1:]] or [5000]:
        for h, U in enumerate(cost_matrices(n)):
            M = HostMatrix(U)
            print("N = %d, hypothesis %s" % (n, ("11 (right frame)", "12 (wrong frame)")[h]), flush=True)
            info = {}
            t = time.perf_counter()
            L.solve_core(M, info)
            print("  all-host flow (the product): %d auction bids, %d Dijkstra steps, %d augmentations, solve %.1f ms on this host; pricing rounds %s"
                  % (info["auction_bids"], info["steps"], info["augmentations"], 1e3 * info["core_seconds"], info["violated_per_round"]), flush=True)
            v0, ecol, ecost, cnt, delta, width = core_of(M)
            stats, log = dict(bids=0, rounds=0, wide_bids=0, width=width), []
            jacobi(ecol, ecost, -v0.copy(), eps_schedule(L.AUCTION["eps0"] * width, width), 0, stats, log)
            print("  Jacobi auction run to completion, first pass (eps %.2g -> %.0e of the core's width):" % (L.AUCTION["eps0"], L.AUCTION["eps_min"]))
            print("\n".join(log))
            print("    total %d rounds, %d bids, %d of them (%.0f %%) in rounds of more than 64 bidders"
                  % (stats["rounds"], stats["bids"], stats["wide_bids"], 100.0 * stats["wide_bids"] / max(stats["bids"], 1)), flush=True)
            for cut, tail in ((0.005, 2.0), (0.005, 16.0), (0.02, 2.0)):
                r = device_like_flow(M, cut, tail)
                print("  device-like flow, phases cut at %.1f %% unassigned, host tail budget %g bids per row: %s" % (100 * cut, tail,
                      {k: (round(x, 4) if isinstance(x, float) else x) for k, x in r.items()}), flush=True)
