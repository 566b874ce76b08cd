"""get_shape_context with NON-DEFAULT binning arguments (shape_context.py:10: r_inner, r_outer, n_rbins, n_thetabins, n_phibins)
on the HIP path — pm_shape_context_neighbors_binned + the host step tables of estimate_transform/binning.py — against the
histograms the unmodified reference returned for the same calls (tests/golden/gen_binning.py -> binning.npz: ten parameter sets x
five neighbour sets x three mean distances; Gaussian clouds, half-integer lattices with neighbours ON sector planes and axes, edge
vectors incl. zero, NaN, +-0 and overflow / underflow): float64 bit patterns must be equal.  No oracle in between."""
import warnings

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sc_mod():
    import torch
    from platymatch_amd import _native as nat
    from platymatch_amd.build import build_native
    build_native()
    nat.load()
    assert torch.cuda.is_available()
    from platymatch_amd.estimate_transform import shape_context
    return shape_context


def same_bits(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))


def test_every_parameter_set_of_the_fixture_gives_the_references_histogram(sc_mod):
    g = load_golden("binning")
    n = 0
    for pi, p in enumerate(g["params"]):
        args = dict(r_inner=float(p[0]), r_outer=float(p[1]), n_rbins=int(p[2]), n_thetabins=int(p[3]), n_phibins=int(p[4]))
        for name in g["set_names"]:
            nb = g["nb_" + str(name)]
            for mi, md in enumerate(g["mean_dists"]):
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    got = sc_mod.get_shape_context(nb, float(md), **args)
                ref = g["sc_p%d_%s_m%d" % (pi, name, mi)]
                assert isinstance(got, np.ndarray) and same_bits(got, ref), (args, str(name), float(md))
                n += 1
    assert n == 150


def test_the_kernel_lists_neighbours_on_azimuth_steps_and_counts_the_rest(sc_mod):
    """The device's share alone: on a generic cloud nothing is left to the host; on the lattice the listed rows are exactly those
    within 2^-46 of a step by NumPy's own arctan2 (or nearer than the two libraries can differ), and counts + listed + dropped
    account for every neighbour."""
    from platymatch_amd import _kernels as K
    from platymatch_amd import _native as nat
    from platymatch_amd.estimate_transform import binning as B
    g = load_golden("binning")
    nt, nph = 5, 8
    edges = B.r_edges(0.1, 3.0, 4)
    for name, expect_listed in (("gauss200", False), ("lattice343", True)):
        nb = g["nb_" + name]
        counts, rows = K.shape_context_neighbors_binned(nat.to_dev(nb), 1.0, edges, B.cos_steps(nt), B.phi_steps(nph), nt, nph)
        assert counts.shape == (4 * nt * nph,) and counts.min() >= 0
        assert (rows.size > 0) == expect_listed
        with np.errstate(all="ignore"):
            phi = np.arctan2(nb[:, 1], nb[:, 0])
            phi = np.where(phi < 0, 2 * np.pi + phi, phi)
            near = np.abs(phi[:, None] - B.phi_steps(nph)[None, :]).min(1)
            r_ = np.sqrt((nb[:, 0] ** 2 + nb[:, 1] ** 2) + nb[:, 2] ** 2)
            valid = np.abs(nb[:, 2] / r_) <= 1.0
        assert set(rows.tolist()) <= set(np.flatnonzero(valid & (near <= 2.0 ** -45)).tolist())
        assert set(np.flatnonzero(valid & (near <= 2.0 ** -47)).tolist()) <= set(rows.tolist())
        # every valid neighbour is counted, listed, or spills past the last bin (theta = pi in the outermost ring)
        assert counts.sum() + rows.size <= int(valid.sum())
        assert int(valid.sum()) - (counts.sum() + rows.size) <= int((valid & (nb[:, 2] < 0) & (nb[:, 0] == 0) & (nb[:, 1] == 0)).sum())


def test_torch_in_torch_out_and_bad_arguments(sc_mod):
    import torch
    from platymatch_amd import _native as nat
    g = load_golden("binning")
    nb = g["nb_gauss200"]
    ref = g["sc_p3_gauss200_m1"]
    p = g["params"][3]
    got = sc_mod.get_shape_context(nat.to_dev(nb), float(g["mean_dists"][1]), float(p[0]), float(p[1]), int(p[2]), int(p[3]), int(p[4]))
    assert torch.is_tensor(got) and got.is_cuda and same_bits(got.cpu().numpy(), ref)
    for bad in (dict(n_rbins=0), dict(n_thetabins=-1), dict(n_phibins=2.5), dict(n_phibins=10 ** 6)):
        with pytest.raises(ValueError):
            sc_mod.get_shape_context(nb, 1.0, **bad)
    with pytest.raises(ValueError):
        sc_mod.get_shape_context(nb[:, :2], 1.0, n_phibins=8)


def test_default_arguments_still_take_the_compiled_tables(sc_mod):
    """The default call and the same binning spelled through the general entry agree with each other and with the reference."""
    g = load_golden("binning")
    from platymatch_amd import _kernels as K
    from platymatch_amd import _native as nat
    from platymatch_amd.estimate_transform import binning as B
    for name in ("gauss200", "lattice343", "edge_vectors"):
        nb = g["nb_" + name]
        ref = g["sc_p0_%s_m0" % name]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            assert same_bits(sc_mod.get_shape_context(nb, 1.0), ref)
        edges = B.r_edges(1 / 8, 2, 5)
        counts, rows = K.shape_context_neighbors_binned(nat.to_dev(nb), 1.0, edges, B.cos_steps(6), B.phi_steps(12), 6, 12)
        idx = B.bin_rows(nb[rows], 1.0, edges, 6, 12)
        np.add.at(counts, idx[idx >= 0], 1)
        with np.errstate(all="ignore"):
            sc = counts.astype(np.float64)
            sc = sc / sc.sum()
        assert same_bits(sc, ref)
