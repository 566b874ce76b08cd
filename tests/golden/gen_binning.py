#!/usr/bin/env python3
"""The unmodified reference's get_shape_context (shape_context.py:10-42) called with NON-DEFAULT binning parameters
(r_inner, r_outer, n_rbins, n_thetabins, n_phibins): the one place of the path where the reference's signature accepts them
(get_unary always passes the defaults).  For every (parameter set, neighbour set, mean distance) the normalised histogram is
stored as float64 together with the inputs.  Neighbour sets: generic Gaussian clouds, half-integer lattices (neighbours ON sector
planes, polar axis, the negative x axis), hand-made edge vectors (zero vector, +-z axis, y = +-0, tiny negative y, NaN).
Data only; build container only.
Usage: python tests/golden/gen_binning.py"""
import contextlib
import io
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402

PARAMS = [            # (r_inner, r_outer, n_rbins, n_thetabins, n_phibins)
    (1 / 8, 2, 5, 6, 12),        # the default, as a control
    (0.1, 3.0, 4, 5, 8),
    (1 / 8, 2, 5, 4, 16),
    (0.2, 1.5, 3, 7, 9),
    (1 / 16, 4, 7, 3, 5),
    (1 / 8, 2, 1, 1, 1),
    (0.5, 1.0, 2, 2, 2),
    (1 / 8, 2, 6, 12, 24),
    (0.3, 0.3, 3, 4, 6),         # r_inner == r_outer: three equal edges
    (2.0, 0.125, 5, 6, 12),      # decreasing edges: "first edge with r < edge" is not a count
]


def neighbour_sets():
    rng = np.random.default_rng(20260405)
    sets = {}
    sets["gauss200"] = rng.normal(size=(200, 3)) * np.array([30.0, 20.0, 12.0])
    sets["gauss57_small"] = rng.normal(size=(57, 3)) * 0.37
    g = np.arange(-3, 4) * 0.5
    lat = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    sets["lattice343"] = lat                              # contains the zero vector, the axes, the diagonals
    sets["lattice_scaled"] = lat[rng.permutation(len(lat))[:150]] * 7.0 + 0.0
    edge = [[0.0, 0.0, 0.0], [0.0, 0.0, 1.0], [0.0, 0.0, -1.0], [1.0, 0.0, 0.0], [-1.0, 0.0, 0.0], [-1.0, -0.0, 0.0],
            [1.0, -0.0, 0.0], [0.0, 1.0, 0.0], [0.0, -1.0, 0.0], [1.0, -1e-30, 0.0], [1.0, -1e-17, 0.0], [1.0, 1e-17, 0.0],
            [1.0, 1.0, 0.0], [-1.0, 1.0, 0.0], [-1.0, -1.0, 0.0], [1.0, -1.0, 0.0], [1.0, 1.0, 1.0], [-1.0, -1.0, -1.0],
            [np.nan, 1.0, 1.0], [1.0, np.nan, 1.0], [1.0, 1.0, np.nan], [3.0, 4.0, 0.0], [0.0, 3.0, 4.0], [3.0, 0.0, -4.0],
            [1e-200, 1e-200, 1e-200], [1e150, 1e150, 1e150], [0.5, 0.8660254037844386, 0.0], [0.8660254037844386, 0.5, 0.0],
            [-0.5, 0.8660254037844386, 1.0], [0.25, 0.0, 0.0], [0.125, 0.0, 0.0], [0.0, 0.5, 0.0], [0.0, 0.0, 2.0]]
    sets["edge_vectors"] = np.array(edge, dtype=np.float64)
    return sets


def main():
    sc_mod = G.import_reference()[0]
    out = {"params": np.array(PARAMS, dtype=np.float64)}
    sets = neighbour_sets()
    for name, nb in sets.items():
        out["nb_" + name] = nb
    out["mean_dists"] = np.array([1.0, 37.3, 0.41])
    out["set_names"] = np.array(sorted(sets))
    n_cases = 0
    for pi, (r_in, r_out, nr, nt, nph) in enumerate(PARAMS):
        for name in sorted(sets):
            for mi, md in enumerate(out["mean_dists"]):
                with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings(), np.errstate(all="ignore"):
                    warnings.simplefilter("ignore")
                    sc = sc_mod.get_shape_context(sets[name], float(md), r_inner=r_in, r_outer=r_out, n_rbins=int(nr),
                                                  n_thetabins=int(nt), n_phibins=int(nph))
                out["sc_p%d_%s_m%d" % (pi, name, mi)] = np.asarray(sc, dtype=np.float64)
                n_cases += 1
    path = os.path.join(HERE, "binning.npz")
    np.savez_compressed(path, **out)
    print("wrote %d cases, %d bytes" % (n_cases, os.path.getsize(path)))


if __name__ == "__main__":
    main()
