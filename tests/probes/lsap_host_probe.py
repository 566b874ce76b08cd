"""CPU probe of the assignment stage's HOST chain (csrc/pm_lsap_core.cpp: auction + shortest augmenting paths) on REAL chi-square
matrices, without a GPU: the matrices of a synthetic pair (bench.synth) come from the oracle's C port (why this probe lives under
tests/: only tests may touch oracle/), the dense-matrix kernels are replaced by the NumPy double of tests/test_lsap_core.py, and
lsap.solve_core + certify run as in the product.  Reports the core's own time per phase (auction / search), its counters, and
whether the certified answer equals SciPy's at sizes where SciPy finishes.

    python tests/probes/lsap_host_probe.py 5000 [20000] [--cache DIR] [--scipy]

The first run at a size builds and caches the four natural-order matrices (U11, U12, U13, U14; the twins share their solves) under
DIR (default /tmp/pm_lsap_probe): ~20 s at 5k, ~6 min at 20k on 8 cores."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def matrices(n, cache):
    os.makedirs(cache, exist_ok=True)
    paths = [os.path.join(cache, "U_%d_%d.npy" % (n, t)) for t in range(4)]
    if all(os.path.exists(p) for p in paths):
        return [np.load(p, mmap_mode="r") for p in paths]
    import oracle
    import bench
    oracle.build()
    oracle.set_threads(oracle.host_threads())
    mv, fx, _ = bench.synth(n)
    t = time.perf_counter()
    cm, cf = oracle.get_centroid(mv, False), oracle.get_centroid(fx, False)
    mdm, mdf = oracle.get_mean_distance(mv, False), oracle.get_mean_distance(fx, False)
    um = oracle.get_unary(cm, mdm, mv, "moving", x0=oracle.pca_axis(mv.T))
    uf = oracle.get_unary(cf, mdf, fx, "fixed", x0=oracle.pca_axis(fx.T))
    print("descriptors %.1f s" % (time.perf_counter() - t), flush=True)
    for b in range(4):
        t = time.perf_counter()
        U = oracle.unary_distance_matrix(um[0], uf[b])
        np.save(paths[b], U)
        print("matrix 1%d: %.1f s" % (b + 1, time.perf_counter() - t), flush=True)
    return [np.load(p, mmap_mode="r") for p in paths]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    cache = "/tmp/pm_lsap_probe"
    if "--cache" in sys.argv:
        cache = sys.argv[sys.argv.index("--cache") + 1]
        args.remove(cache)
    check = "--scipy" in sys.argv
    from test_lsap_core import HostMatrix
    from platymatch_amd import lsap as L
    for n in (int(a) for a in args):
        Us = matrices(n, cache)
        for t, U in enumerate(Us):
            M = HostMatrix(np.asarray(U))
            info = {}
            t0 = time.perf_counter()
            sol = L.solve_core(M, info)
            wall = time.perf_counter() - t0
            ok = sol is not None and L.certify(M, *sol, info=info)
            line = ("n = %d, matrix 1%d: auction %.1f ms (%d bids), search %.1f ms (%d steps, %d augmentations), pricing rounds %s, "
                    "certified unique %s" % (n, t + 1, 1e3 * info.get("auction_seconds", 0.0), info.get("auction_bids", 0),
                                             1e3 * info.get("core_seconds", 0.0), info.get("steps", 0), info.get("augmentations", 0),
                                             info.get("rounds"), ok))
            if check and ok:
                from scipy.optimize import linear_sum_assignment
                line += ", equals SciPy %s" % bool(np.array_equal(linear_sum_assignment(np.asarray(U))[1], sol[2]))
            print(line + "   [wall with the NumPy double %.1f s]" % wall, flush=True)


if __name__ == "__main__":
    main()
