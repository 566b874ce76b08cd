"""pm_lsap_solve (SciPy's rectangular LSA solver restated in C++, callable from threads) against SciPy itself and the
reference fixtures' assignment vectors.  CPU only."""
import numpy as np
import pytest
from scipy.optimize import linear_sum_assignment as scipy_lsa

from conftest import SCENARIOS, load_golden
from platymatch_amd import lsap


def same(M):
    r0, c0 = scipy_lsa(M)
    r1, c1 = lsap.linear_sum_assignment(M)
    return np.array_equal(r0, r1) and np.array_equal(c0, c1) and r1.dtype == np.int64


def test_identical_to_scipy_on_random_tied_rectangular_matrices():
    rng = np.random.default_rng(0)
    for trial in range(4000):
        nr, nc = rng.integers(1, 48, size=2)
        kind = trial % 6
        if kind == 0:
            M = rng.random((nr, nc))
        elif kind == 1:
            M = rng.integers(0, 4, size=(nr, nc)).astype(float)                 # heavy ties
        elif kind == 2:
            M = np.full((nr, nc), 3.0)                                          # constant: identity expected
        elif kind == 3:
            M = rng.integers(0, 2, size=(nr, nc)) * rng.random((nr, nc))          # many exact zeros
        elif kind == 4:
            M = np.round(rng.normal(size=(nr, nc)), 1)                           # negative entries, ties
        else:
            M = rng.random((nr, nc))
            M[rng.random((nr, nc)) < 0.2] = np.inf                               # forbidden pairs
            try:
                scipy_lsa(M)
            except ValueError:
                with pytest.raises(ValueError):
                    lsap.linear_sum_assignment(M)
                continue
        assert same(M), (trial, nr, nc)
    assert same(rng.random((600, 600))) and same(rng.random((300, 700))) and same(rng.random((700, 300)))


def test_scalar_and_vector_paths_agree_with_scipy(monkeypatch):
    """pm_lsap_solve has SciPy's scalar scan and an AVX-512 scan (column order + tie replay): both must be SciPy's
    answer, also where ties are everywhere and the tie replay decides nearly every step."""
    rng = np.random.default_rng(11)
    mats = [rng.integers(0, 3, size=(97, 131)).astype(float), rng.integers(0, 5, size=(200, 200)).astype(float),
            np.round(rng.random((150, 90)), 2), rng.random((257, 263)), np.zeros((33, 65)),
            np.where(rng.random((120, 120)) < 0.3, np.inf, rng.integers(0, 4, size=(120, 120)).astype(float))]
    for M in mats:
        try:
            want = scipy_lsa(M)
        except ValueError:
            want = None
        for scalar in ("1", "0"):
            monkeypatch.setenv("PM_LSAP_SCALAR", scalar)
            if want is None:
                with pytest.raises(ValueError):
                    lsap.linear_sum_assignment(M)
            else:
                got = lsap.linear_sum_assignment(M)
                assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), (M.shape, scalar)


def test_error_behaviour_like_scipy():
    for bad in (np.array([[1.0, np.nan], [0.0, 1.0]]), np.array([[1.0, -np.inf], [0.0, 1.0]])):
        with pytest.raises(ValueError):
            scipy_lsa(bad)
        with pytest.raises(ValueError, match="invalid numeric"):
            lsap.linear_sum_assignment(bad)
    inf = np.array([[np.inf, np.inf], [1.0, 2.0]])
    with pytest.raises(ValueError):
        scipy_lsa(inf)
    with pytest.raises(ValueError, match="infeasible"):
        lsap.linear_sum_assignment(inf)
    with pytest.raises(ValueError):
        lsap.linear_sum_assignment(np.zeros(3))
    r, c = lsap.linear_sum_assignment(np.zeros((0, 5)))
    assert r.size == 0 and c.size == 0


@pytest.mark.parametrize("name", SCENARIOS)
def test_reference_fixture_assignments(oracle, name):
    """The eight assignment vectors the reference produced (scipy, in the generation run) from the bit-exact cost matrices."""
    d = load_golden(name)
    um = oracle.normalise_counts(*oracle.shape_context_counts(d["centroid_m"], d["mean_dist_m"], d["moving"], "moving", x0=d["x0_m"]))
    uf = oracle.normalise_counts(*oracle.shape_context_counts(d["centroid_f"], d["mean_dist_f"], d["fixed"], "fixed", x0=d["x0_f"]))
    U = [oracle.unary_distance_matrix(um[int(h[0]) - 1], uf[int(h[1]) - 1]) for h in oracle.HYPOTHESES]
    out = lsap.solve_many(U, threads=4)                                     # threaded == sequential == scipy
    for h in range(8):
        assert np.array_equal(out[h][0], d["lsa_rows"][h]) and np.array_equal(out[h][1], d["lsa_cols"][h])
    seq = lsap.solve_many(U, threads=1)
    assert all(np.array_equal(a[1], b[1]) for a, b in zip(out, seq))
