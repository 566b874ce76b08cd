"""The sparse-core assignment solve (platymatch_amd/lsap.py: solve_core + certify, csrc/pm_lsap_core.cpp) against
scipy.optimize.linear_sum_assignment — the widget's call (_dock_widget.py:604-611) — on the CPU: the two device kernels
that read the dense matrix are replaced by a NumPy double with the same contract (tests/test_gpu_lsap.py runs the real
kernels), so what is exercised here is the host solver, the pricing loop, the rectangular (dummy row) handling and the
uniqueness certificate: whenever the certificate says "unique", the indices must be SciPy's; tied matrices must be
refused (they go to pm_lsap_solve, SciPy's algorithm step for step)."""
import numpy as np
import pytest
from scipy.optimize import linear_sum_assignment as scipy_lsa

from conftest import load_golden


class HostMatrix:
    """NumPy restatement of DeviceMatrix's contract (pm_lsap_row_select / pm_lsap_certificate)."""

    def __init__(self, U):
        self.U = np.ascontiguousarray(U, dtype=np.float64)
        self.shape = self.U.shape
        self.passes = 0

    def row_select(self, v, k):
        self.passes += 1
        nr, nc = self.U.shape
        red = self.U if v is None else self.U - v[None, :]
        cols = np.full((nr, k), -1, dtype=np.int32)
        costs = np.full((nr, k), np.inf)
        pad = (-nc) % 256
        R = np.concatenate([red, np.full((nr, pad), np.inf)], axis=1).reshape(nr, -1, 256)      # [nr, chunk, class]
        first = R.argmin(axis=1)                                                                 # first minimum per class
        cand_col = first * 256 + np.arange(256)[None, :]
        cand_red = np.take_along_axis(R, first[:, None, :], axis=1)[:, 0, :]
        cand_col = np.where(np.isfinite(cand_red), cand_col, np.iinfo(np.int32).max)
        order = np.lexsort((cand_col, cand_red), axis=1)[:, :k]
        kk = order.shape[1]
        sel = np.take_along_axis(cand_col, order, axis=1)
        ok = sel < nc
        cols[:, :kk] = np.where(ok, sel, -1)
        costs[:, :kk] = np.where(ok, self.U[np.arange(nr)[:, None], np.minimum(sel, nc - 1)], np.inf)
        return cols, costs, int(not np.isfinite(self.U).all())

    def diagonal(self, n):
        return self.block_diagonal(0)[:n]

    def block_diagonal(self, row0):
        r = np.arange(max(0, min(self.U.shape[0], self.U.shape[1] - row0)))
        return self.U[r, r + row0].copy()

    def col_min(self):
        self.passes += 1
        return self.U.min(axis=0)

    def bid(self, v, rows):
        self.passes += 1
        rows = np.asarray(rows, dtype=np.int64)
        if rows.size == 0:
            return np.zeros(0, np.int32), np.zeros(0), np.zeros(0)
        red = self.U[rows] - v[None, :]
        j1 = red.argmin(axis=1)                                   # first minimum = lowest column on ties
        u1 = red[np.arange(rows.size), j1]
        masked = red.copy()
        masked[np.arange(rows.size), j1] = np.inf
        u2 = masked.min(axis=1) if red.shape[1] > 1 else np.full(rows.size, np.inf)
        return j1.astype(np.int32), u1, u2

    def entries(self, rows, cols):
        return self.U[np.asarray(rows, dtype=np.int64), np.asarray(cols, dtype=np.int64)]

    def certificate(self, u, v, col4row, delta, eps, cap):
        self.passes += 1
        red = (self.U - v[None, :]) - u[:, None]
        nr = self.U.shape[0]
        matched = np.zeros(self.U.shape, dtype=bool)
        matched[np.arange(nr), col4row] = True
        viol = int((~(red >= -delta) & ~matched).sum())
        loose = int((~(np.abs(red) <= delta) & matched).sum())
        sel = (red <= eps) & (red >= -delta) & ~matched
        t = np.argwhere(sel).astype(np.int32)
        neg = np.where(matched, 0.0, np.maximum(-red, 0.0))
        neg = np.where(np.isnan(neg), 0.0, neg)
        bound = float(np.abs(red[matched]).sum() + neg.max(axis=1).sum())
        if len(t) > cap:
            return viol, loose, None, None, bound
        return viol, loose, t, red[sel], bound


def run(U, force_k=None):
    from platymatch_amd import lsap as L
    n, m = U.shape
    M = HostMatrix(U if n <= m else U.T)
    info = {}
    sol = L.solve_core(M, info)
    if sol is None:
        return None, info
    ok = L.certify(M, *sol, info=info)
    return (L._answer(sol[2], n, m) if ok else None), info


@pytest.mark.parametrize("shape", [(40, 40), (300, 300), (257, 300), (300, 257), (64, 900), (700, 90), (1, 5), (5, 1), (600, 640)])
def test_generic_matrices_are_certified_and_equal_scipy(shape):
    rng = np.random.default_rng(sum(shape))
    for trial in range(4):
        U = rng.random(shape) if trial % 2 == 0 else rng.random(shape) * rng.random((1, shape[1])) + 0.3 * rng.random((shape[0], 1))
        got, info = run(U)
        assert got is not None, info
        r, c = scipy_lsa(U)
        assert np.array_equal(got[0], r) and np.array_equal(got[1], c), info


@pytest.mark.parametrize("shape", [(300, 300), (257, 300), (64, 900)])
def test_row_reduction_warm_start_gives_the_same_certified_answer(shape):
    """lsap.ROW_REDUCTION_ROUNDS > 0 (off by default): the bidding rounds change the starting duals and matching, never the result."""
    from platymatch_amd import lsap as L
    rng = np.random.default_rng(sum(shape) + 5)
    U = rng.random(shape)
    old = L.ROW_REDUCTION_ROUNDS
    try:
        L.ROW_REDUCTION_ROUNDS = 12
        got, info = run(U)
    finally:
        L.ROW_REDUCTION_ROUNDS = old
    assert got is not None, info
    r, c = scipy_lsa(U)
    assert np.array_equal(got[0], r) and np.array_equal(got[1], c)


def test_structured_costs_need_pricing_rounds_and_still_equal_scipy():
    """A matrix whose optimum avoids most rows' cheapest entries: decoy columns that are cheap for every row."""
    rng = np.random.default_rng(3)
    n = 700
    U = rng.random((n, n)) + 1.0
    U[:, :6] = rng.random((n, 6)) * 1e-3                    # six columns everybody wants
    U[np.arange(n), rng.permutation(n)] -= 0.9              # the hidden good matching
    got, info = run(U)
    assert got is not None and info["rounds"] >= 2, info
    r, c = scipy_lsa(U)
    assert np.array_equal(got[1], c)


@pytest.mark.parametrize("name", ["insitu02_affine", "insitu04_affine", "synth96x128", "synth1000"])
def test_reference_cost_matrices(oracle, name):
    """The eight matrices of a reference scenario (rebuilt by the oracle from the fixture's histograms): certified, equal to the
    assignment vectors the reference's own SciPy calls produced, and each twin accepts its sibling's duals."""
    from platymatch_amd import lsap as L
    d = load_golden(name)
    um = [d["counts_m%d" % k].astype(np.float64) / d["total_m%d" % k][:, None] for k in (1, 2)]
    uf = [d["counts_f%d" % k].astype(np.float64) / d["total_f%d" % k][:, None] for k in (1, 2, 3, 4)]
    if name == "synth1000":
        um, uf = [a[:400] for a in um], [b[:450] for b in uf]
    Us = [oracle.unary_distance_matrix(um[int(h[0]) - 1], uf[int(h[1]) - 1]) for h in oracle.HYPOTHESES]
    sols = {}
    for h in range(8):
        n, m = Us[h].shape
        M = HostMatrix(Us[h] if n <= m else Us[h].T)
        sol = L.solve_core(M)
        assert sol is not None and L.certify(M, *sol)
        sols[h] = sol
        r, c = scipy_lsa(Us[h])
        got = L._answer(sol[2], n, m)
        assert np.array_equal(got[0], r) and np.array_equal(got[1], c), h
        if name != "synth1000":
            assert np.array_equal(c, d["lsa_cols"][h])
    for twin, h in L.TWINS.items():
        n, m = Us[twin].shape
        M = HostMatrix(Us[twin] if n <= m else Us[twin].T)
        assert L.certify(M, *sols[h])                        # one solve serves both, proven on the twin's own entries
        assert np.array_equal(sols[h][2], sols[twin][2])


def test_tied_matrices_are_refused_not_guessed():
    from platymatch_amd import lsap as L
    rng = np.random.default_rng(0)
    refused = 0
    for trial in range(40):
        n, m = int(rng.integers(2, 40)), int(rng.integers(2, 40))
        U = rng.integers(0, 4, size=(n, m)).astype(np.float64)          # few distinct values: many optimal assignments
        got, info = run(U)
        if got is None:
            refused += 1
        else:                                                            # certified unique: then it must be SciPy's answer
            r, c = scipy_lsa(U)
            assert np.array_equal(got[1], c), (trial, info)
    assert refused >= 30
    # duplicate rows (duplicate nuclei give identical descriptor rows): two optimal assignments -> refused
    U = rng.random((30, 30))
    U[7] = U[3]
    assert run(U)[0] is None
    # constant matrix, and a duplicate COLUMN in a wide matrix
    assert run(np.ones((6, 6)))[0] is None
    W = rng.random((10, 14))
    r, c = scipy_lsa(W)
    W2 = np.concatenate([W, W[:, c[:1]]], axis=1)                       # a free column identical to a used one
    assert run(W2)[0] is None


def test_non_finite_entries_are_left_to_the_dense_solver():
    U = np.random.default_rng(1).random((20, 20))
    U[3, 4] = np.nan
    assert run(U)[0] is None
    U[3, 4] = np.inf
    assert run(U)[0] is None


def test_certificate_rejects_a_wrong_assignment_and_perturbed_duals():
    from platymatch_amd import lsap as L
    U = np.random.default_rng(5).random((50, 60))
    M = HostMatrix(U)
    u, v, c4r = L.solve_core(M)
    assert L.certify(M, u, v, c4r)
    bad = c4r.copy()
    bad[[0, 1]] = bad[[1, 0]]
    assert not L.certify(M, u, v, bad)
    assert not L.certify(M, u + 1e-6, v, c4r)
    v2 = v.copy()
    free = np.setdiff1d(np.arange(60), c4r)
    v2[free[0]] -= 0.5                                    # a free column priced below the matched ones: not a rectangular optimum
    info = {}
    assert not L.certify(M, u, v2, c4r, info=info) and info["optimal"] is False         # not a near-tie: no optimality claim


def test_engineered_near_ties_are_certified_only_above_the_margin():
    """A unique optimum with an alternative exactly `gap` more expensive: two rows' cross entries are set, from the optimal
    duals, to reduced cost gap/2 each, so that swapping their columns costs +gap and nothing else changes.  Above the margin
    the route certifies the optimum (and it is SciPy's answer); below it, it must refuse — a refusal is answered by SciPy's
    own algorithm in the product — and a certified answer must never differ from SciPy's."""
    from platymatch_amd import lsap as L
    rng = np.random.default_rng(8)
    n = 120
    base = rng.random((n, n)) + 0.5
    M = HostMatrix(base)
    u, v, c = L.solve_core(M)
    assert L.certify(M, u, v, c) and np.array_equal(c, scipy_lsa(base)[1])
    i1, i2 = 3, 77
    certified = {}
    for gap in (1e-3, 1e-6, 1e-9, 1e-10, 1e-12, 1e-14, 0.0):
        U = base.copy()
        U[i1, c[i2]] = (u[i1] + v[c[i2]]) + gap / 2
        U[i2, c[i1]] = (u[i2] + v[c[i1]]) + gap / 2
        got, info = run(U)
        rs, cs = scipy_lsa(U)
        if gap >= 1e-9:
            assert np.array_equal(cs, c)                                       # the engineered alternative really is worse
        certified[gap] = got is not None
        if got is not None:
            assert np.array_equal(got[1], cs), (gap, info)
        else:
            # refused for uniqueness only: the certificate still says OPTIMAL (what accept_near_ties builds on), and the
            # solver's assignment indeed costs what SciPy's does, to rounding
            assert info["optimal"] and info.get("unique") is False, info
            sol = L.solve_core(HostMatrix(U))
            assert abs(U[np.arange(n), sol[2]].sum() - U[rs, cs].sum()) <= 1e-12 * n
    assert certified[1e-3] and certified[1e-6] and certified[1e-9]
    assert not certified[1e-14] and not certified[0.0]                        # within the margin (floor 1e-11 of the scale): refused


@pytest.mark.parametrize("shape", [(400, 400), (990, 1000), (257, 300), (64, 900), (1200, 1200)])
def test_auction_warm_start_never_changes_the_certified_answer(shape, monkeypatch):
    """lsap.AUCTION (eps-scaling forward auction over the core before the first shortest-path solve; reverse steps for spare
    columns, searches from the column side for the columns it strands): off, on, on for near-square problems only, starved of bids (the budget stops it mid-round) and with a
    coarse final eps — the certified assignment is SciPy's every time; only the amount of search left differs."""
    from platymatch_amd import lsap as L
    rng = np.random.default_rng(shape[0] * 7 + shape[1])
    U = rng.random(shape) * rng.random((1, shape[1])) + 0.3 * rng.random((shape[0], 1)) + 0.05 * rng.random(shape)
    r, c = scipy_lsa(U)
    base = dict(L.AUCTION)
    steps = {}
    for name, setting in (("off", None), ("on", base), ("near-square only", dict(base, max_free_columns=0.02)),
                          ("starved", dict(base, max_free_columns=1.0, bids_per_row=1)),
                          ("coarse", dict(base, max_free_columns=1.0, eps_min=0.05, rounds=1))):
        monkeypatch.setattr(L, "AUCTION", setting)
        got, info = run(U)
        assert got is not None, (name, info)
        assert np.array_equal(got[0], r) and np.array_equal(got[1], c), (name, info)
        steps[name] = info["steps"]
        assert ("auction_bids" in info) == (name != "off" and (name != "near-square only" or shape[1] - shape[0] <= 0.02 * shape[1])), (name, info)
    if shape[0] == shape[1]:
        assert steps["on"] < steps["off"], steps                     # the point of it: less search left


def test_rectangular_cores_do_not_scan_dummy_rows_densely():
    """With spare columns the squared problem's dummy rows hold columns too; a search must not pay a dense scan for each one
    it meets (pm_lsap_core.cpp: scan_row), nor a phase spoil the duals of spare columns.  The counters say so: at most a few
    dummy scans per dummy row."""
    from platymatch_amd import lsap as L
    rng = np.random.default_rng(77)
    n, m = 700, 1000
    U = rng.random((n, m)) * rng.random((1, m)) + 0.2 * rng.random((n, 1))
    got, info = run(U)
    assert got is not None and np.array_equal(got[1], scipy_lsa(U)[1])
    dummy_scans = info["dummy_scans"] % 1000000
    assert dummy_scans <= 3 * (m - n), info


def test_fuzz_of_shapes_and_cost_distributions_against_scipy():
    """Random shapes (square, wide, tall) and cost distributions — uniform, column-scaled, heavy-tailed, squared distances of
    point sets, negative entries, costs on a coarse grid plus noise (near-ties) — through solve_core + certify with the
    auction warm start on: whatever is certified equals SciPy's answer; what is not certified is rare."""
    rng = np.random.default_rng(2024)
    uncertified = 0
    for it in range(70):
        n, m = int(rng.integers(2, 500)), int(rng.integers(2, 500))
        kind = it % 6
        if kind == 0:
            U = rng.random((n, m))
        elif kind == 1:
            U = rng.random((n, m)) * rng.random((1, m)) + 0.3 * rng.random((n, 1))
        elif kind == 2:
            U = np.abs(rng.normal(size=(n, m))) ** 3
        elif kind == 3:
            a, b = rng.random((n, 3)), rng.random((m, 3))
            U = ((a[:, None, :] - b[None, :, :]) ** 2).sum(-1)
        elif kind == 4:
            U = rng.random((n, m)) - 0.5
        else:
            U = np.round(rng.random((n, m)) * 50) / 50 + 1e-9 * rng.random((n, m))
        got, info = run(U)
        if got is None:
            uncertified += 1
            continue
        r, c = scipy_lsa(U)
        assert np.array_equal(got[0], r) and np.array_equal(got[1], c), (it, n, m, kind, info)
    assert uncertified <= 3


def test_near_ties_are_settled_on_their_blocks_by_the_dense_algorithm():
    """lsap.resolve_near_ties (round 4): a certified optimum with alternatives inside the margin — here two engineered 2-cycles and
    a 3-cycle, each worth +-1e-14 — is not refused: the rows each cycle connects are assigned by SciPy's algorithm on their own
    block (2 x 2, 2 x 2, 3 x 3 entries fetched through M.entries) and spliced in.  The answer is SciPy's on the whole matrix."""
    from platymatch_amd import lsap as L
    rng = np.random.default_rng(31)
    n = 200
    base = rng.random((n, n)) + 0.5
    u, v, c = L.solve_core(HostMatrix(base))
    for sign in (+1.0, -1.0):
        U = base.copy()
        for cyc in ((3, 77), (10, 150), (20, 60, 120)):
            k = len(cyc)
            for a in range(k):                               # row cyc[a] may take the column of cyc[a + 1]: an alternating cycle
                i, j = cyc[a], c[cyc[(a + 1) % k]]
                U[i, j] = (u[i] + v[j]) + sign * 1e-14 / k
        M = HostMatrix(U)
        info = {}
        sol = L.solve_core(M, info)
        assert not L.certify(M, *sol, info=info) and info["optimal"]
        got = L.resolve_near_ties(M, sol, info)
        assert got is not None and sorted(info["resolved_groups"]) == [2, 2, 3]
        assert np.array_equal(got, scipy_lsa(U)[1]), sign
    # spare columns (nr < nc): a 2-cycle among rows that hold columns is settled the same way — unless an alternative could also run
    # through a column nobody holds (near-tight entries into one AND held columns priced like a free one), which is left alone
    R = rng.random((40, 50)) + 0.5
    u, v, c = L.solve_core(HostMatrix(R))
    R2 = R.copy()
    R2[7, c[21]] = (u[7] + v[c[21]]) - 0.5e-14
    R2[21, c[7]] = (u[21] + v[c[7]]) - 0.5e-14
    M = HostMatrix(R2)
    info = {}
    sol = L.solve_core(M, info)
    assert not L.certify(M, *sol, info=info) and info["optimal"]
    got = L.resolve_near_ties(M, sol, info)
    assert got is not None and 2 in info["resolved_groups"] and np.array_equal(got, scipy_lsa(R2)[1])
    # round 5: an alternative THROUGH a spare column — a row whose own column is priced like a free one (a dummy row could take it
    # at no cost) is also near-tight on a column nobody holds: releasing one and taking the other is an alternating path, a cycle
    # through the digraph's node F.  The row gets a rectangular block (its column + the free column) and SciPy's algorithm decides.
    settled = 0
    for seed in range(40):
        rg = np.random.default_rng(500 + seed)
        R = rg.random((40, 50)) + 0.5
        u, v, c = L.solve_core(HostMatrix(R))
        free = np.setdiff1d(np.arange(50), c)
        v_free = v[free].min()
        cand = [i for i in range(40) if v[c[i]] == v_free]          # rows holding a column at the free columns' level
        if not cand:
            continue
        for sign in (+1.0, -1.0):
            R3 = R.copy()
            i, f = cand[0], int(free[seed % len(free)])
            R3[i, f] = (u[i] + v[f]) + sign * 1e-14                  # row i is indifferent (to 1e-14) between its column and the free one
            M = HostMatrix(R3)
            info = {}
            sol = L.solve_core(M, info)
            if sol is None or L.certify(M, *sol, info=info) or not info.get("optimal"):
                continue
            got = L.resolve_near_ties(M, sol, info)
            assert got is not None, (seed, sign, info)
            assert np.array_equal(got, scipy_lsa(R3)[1]), (seed, sign)
            settled += int(info.get("resolved_through_spare_columns", 0) > 0)
    assert settled >= 5                                              # the spare-column route was really exercised


@pytest.mark.parametrize("shape", [(300, 300), (257, 300), (300, 257), (900, 900)])
def test_a_solve_on_a_perturbed_matrix_is_certified_against_the_exact_matrix_on_its_listed_entries(shape):
    """lsap.certify_listed — what cost_mode='relaxed' rests on: the assignment is solved on R (every entry within cost_delta
    of the exact matrix C), then proven to be C's unique optimum from C's values at the matched and the near-tight entries
    alone (C is never read elsewhere: the entry function counts what it is asked for).  The verdict must be SciPy's
    answer on C whenever it is given."""
    from platymatch_amd import lsap as L
    rng = np.random.default_rng(shape[0] * 3 + shape[1])
    n, m = shape
    C = rng.random(shape) * rng.random((1, m)) + 0.3 * rng.random((n, 1)) + 0.05 * rng.random(shape)
    cost_delta = 1e-11                                             # (a hundred times the solver's own tolerance)
    R = C + rng.uniform(-cost_delta, cost_delta, size=shape)
    T = (lambda X: X if n <= m else X.T)
    M = HostMatrix(T(R))
    sol = L.solve_core(M)
    assert sol is not None
    asked = []

    def entries(rows, cols):
        asked.append(len(rows))
        return (T(C)[rows, cols], T(C)[rows, cols] + 0.0)

    infos = [{}, {}]
    ok = L.certify_listed(M, *sol, exact_entries=entries, cost_delta=cost_delta, infos=infos)
    assert ok == [True, True], infos
    assert asked and asked[0] < 12 * max(n, m)                     # a few entries per row, not the matrix
    got = L._answer(sol[2], n, m)
    r, c = scipy_lsa(C)
    assert np.array_equal(got[0], r) and np.array_equal(got[1], c)
    # the premise is checked where C is known: an entry function that contradicts cost_delta is refused
    bad = L.certify_listed(M, *sol, exact_entries=lambda rows, cols: (T(C)[rows, cols] + 1e-9 * (np.arange(len(rows)) == 5),),
                           cost_delta=cost_delta, infos=[{}])
    assert bad == [False]


def test_listed_certificate_refuses_what_the_perturbation_decided():
    """C has two optima `gap` apart (a 2-cycle, engineered from the optimal duals as above).  The perturbed matrix R breaks the
    tie one way or the other and its own certificate would call the result unique; certified against C's listed entries the
    answer is refused when gap is inside the exact mode's margin, refused when R picked C's loser, and accepted — and SciPy's —
    when the gap is well above the margin and R picked the winner."""
    from platymatch_amd import lsap as L
    rng = np.random.default_rng(21)
    n = 150
    base = rng.random((n, n)) + 0.5
    u, v, c = L.solve_core(HostMatrix(base))
    i1, i2 = 5, 90
    cost_delta = 1e-9
    for gap, push in ((0.0, +1), (1e-14, -1), (4e-10, -1), (4e-10, +1), (1e-6, +1)):
        C = base.copy()
        C[i1, c[i2]] = (u[i1] + v[c[i2]]) + gap / 2
        C[i2, c[i1]] = (u[i2] + v[c[i1]]) + gap / 2
        R = C.copy()
        R[i1, c[i2]] += push * 0.9 * cost_delta          # push = -1: the perturbation makes C's loser (the swap) the cheaper one in R
        R[i2, c[i1]] += push * 0.9 * cost_delta
        M = HostMatrix(R)
        sol = L.solve_core(M)
        info = {}
        ok = L.certify_listed(M, *sol, exact_entries=lambda rows, cols: (C[rows, cols],), cost_delta=cost_delta, infos=[info])[0]
        rs, cs = scipy_lsa(C)
        swapped = not np.array_equal(sol[2], c)
        if gap <= 1e-14:
            assert not ok, (gap, push, info)                          # a tie of the exact matrix: never certified
        elif push < 0:
            assert swapped and not ok, (gap, push, info)              # R chose the assignment that is NOT C's optimum
        else:
            assert ok and np.array_equal(sol[2], cs), (gap, push, info)


@pytest.mark.parametrize("shape", [(300, 300), (257, 300), (300, 257), (1000, 1000)])
def test_an_approximate_matrix_as_a_filter_gives_the_exact_matrices_certified_answer(shape):
    """lsap.FilteredMatrix: the dense matrix the solver queries is only within cost_delta = 1e-6 of the exact one (what a float32
    cost build delivers); every cost that reaches the sparse core or the certificate is evaluated exactly on request.  The
    certified answer must be SciPy's on the EXACT matrix although the approximate matrix's own optimum is usually another
    assignment; the exact matrix is only ever read at listed entries (a few dozen per row)."""
    from platymatch_amd import lsap as L
    rng = np.random.default_rng(shape[0] * 5 + shape[1])
    n, m = shape
    C = rng.random(shape) * rng.random((1, m)) + 0.3 * rng.random((n, 1)) + 0.05 * rng.random(shape)
    Ct = C + rng.uniform(-4e-14, 4e-14, size=shape)              # the twin: the same terms in another order
    cost_delta = 1e-6
    A = C + rng.uniform(-cost_delta, cost_delta, size=shape)
    T = (lambda X: X if n <= m else X.T)
    asked = []

    def entries(rows, cols):
        asked.append(len(rows))
        return T(C)[rows, cols], T(Ct)[rows, cols]

    M = L.FilteredMatrix(HostMatrix(T(A)), entries, cost_delta)
    info = {}
    sol = L.solve_core(M, info)
    assert sol is not None, info
    infos = [{}, {}]
    ok = L.certify_listed(M, *sol, exact_entries=entries, cost_delta=cost_delta, infos=infos)
    assert ok == [True, True], (info, infos)
    got = L._answer(sol[2], n, m)
    r, c = scipy_lsa(C)
    assert np.array_equal(got[0], r) and np.array_equal(got[1], c)
    assert np.array_equal(scipy_lsa(Ct)[1], c)
    ra, ca = scipy_lsa(A)
    print("%s: approximate matrix's own optimum differs from the exact one's in %d rows; exact entries evaluated: %d of %d; polishing rounds %s"
          % (shape, int((ca != c).sum()), sum(asked), n * m, info.get("polish_violated")))
    assert sum(asked) < 0.5 * n * m


def test_filter_settles_what_lies_below_its_own_accuracy_and_refuses_exact_ties():
    """Two optima of the exact matrix `gap` apart with gap far BELOW the filter's accuracy (1e-9 against 1e-6): the approximate
    matrix cannot tell them apart, the exact costs on the listed entries do — the certified answer is SciPy's; gap = 0 (an
    exact tie) is refused."""
    from platymatch_amd import lsap as L
    rng = np.random.default_rng(33)
    n = 200
    base = rng.random((n, n)) + 0.5
    u, v, c = L.solve_core(HostMatrix(base))
    i1, i2 = 7, 120
    cost_delta = 1e-6
    for gap in (1e-9, 0.0):
        C = base.copy()
        C[i1, c[i2]] = (u[i1] + v[c[i2]]) + gap / 2
        C[i2, c[i1]] = (u[i2] + v[c[i1]]) + gap / 2
        A = C + rng.uniform(-cost_delta, cost_delta, size=C.shape)
        M = L.FilteredMatrix(HostMatrix(A), lambda rows, cols: (C[rows, cols],), cost_delta)
        info = {}
        sol = L.solve_core(M, info)
        assert sol is not None
        ci = {}
        ok = L.certify_listed(M, *sol, exact_entries=M.exact_entries, cost_delta=cost_delta, infos=[ci])[0]
        if gap > 0:
            assert ok and np.array_equal(sol[2], scipy_lsa(C)[1]), (gap, info, ci)
        else:
            assert not ok and ci.get("unique") is False, (gap, ci)


def test_fuzz_a_certified_answer_of_the_listed_certificate_is_the_exact_matrices_unique_optimum():
    """Soundness of what cost_mode='relaxed' / 'filter' rest on, by brute force on small problems: whenever certify_listed says yes
    — after a plain solve on the approximate matrix (the relaxed flow) or after the FilteredMatrix flow — the assignment is
    SciPy's on the EXACT matrix and no other assignment comes within 1e-12 of its cost (checked by forbidding each matched entry
    in turn).  Generic, rectangular, duplicated-row (tied) and engineered near-tie matrices; perturbations from 1e-12 to 1e-4."""
    from platymatch_amd import lsap as L
    rng = np.random.default_rng(2024)
    said_yes = {"relaxed": 0, "filter": 0}
    generic = {"relaxed": [0, 0], "filter": [0, 0]}
    for case in range(140):
        n = int(rng.integers(8, 48))
        m = n if case % 3 else n + int(rng.integers(1, 9))
        C = rng.random((n, m)) * rng.random((1, m)) + 0.2 * rng.random((n, 1)) + 0.05 * rng.random((n, m))
        kind = case % 5
        if kind == 3:                                            # tied: a duplicated row
            C[int(rng.integers(0, n))] = C[int(rng.integers(0, n))]
        elif kind == 4:                                          # an alternative `gap` above the optimum
            r, c = scipy_lsa(C)
            i1, i2 = rng.choice(n, size=2, replace=False)
            gap = 10.0 ** rng.uniform(-14, -6)
            C[i1, c[i2]] += (C[i1, c[i1]] + C[i2, c[i2]] + gap) - (C[i1, c[i2]] + C[i2, c[i1]])
        cost_delta = 10.0 ** rng.uniform(-12, -4)
        A = C + rng.uniform(-cost_delta, cost_delta, size=C.shape)

        def entries(rows, cols):
            return (C[rows, cols],)
        for flow in ("relaxed", "filter"):
            M = HostMatrix(A) if flow == "relaxed" else L.FilteredMatrix(HostMatrix(A), entries, cost_delta)
            sol = L.solve_core(M)
            if sol is None:
                continue
            ok = L.certify_listed(M, *sol, exact_entries=entries, cost_delta=cost_delta, infos=[{}])[0]
            if kind < 3:
                generic[flow][0] += 1
                generic[flow][1] += bool(ok)
            if not ok:
                continue
            said_yes[flow] += 1
            r, c = scipy_lsa(C)
            assert np.array_equal(sol[2], c), (case, flow, kind, cost_delta)
            best = C[r, c].sum()
            for i in range(n):                                   # uniqueness by brute force
                D = C.copy()
                D[i, c[i]] = 1e6
                r2, c2 = scipy_lsa(D)
                assert D[r2, c2].sum() > best + 1e-12, (case, flow, kind, i, D[r2, c2].sum() - best)
    assert said_yes["relaxed"] > 20 and said_yes["filter"] > 60, said_yes
    assert generic["filter"][1] >= 0.9 * generic["filter"][0], generic        # the filter flow settles generic matrices whatever its accuracy


@pytest.mark.parametrize("mode", ["force", "auto", "0"])
def test_column_side_repair_of_a_pricing_round_never_changes_the_certified_answer(mode, monkeypatch):
    """pm_lsap_core_reprice, round 5: when the matching is complete and the violated rows meet on few columns (a starved core with a
    handful of globally cheap columns: nearly every row undercuts them after the first solve — the shape of the 48 179-row pricing
    round measured at 50 000 nuclei), the columns' duals are lowered and only their holders freed, instead of every violated row.
    Forced for every complete matching, by its criterion, or switched off: the certified answer is SciPy's each time, on square and
    rectangular problems."""
    from platymatch_amd import lsap as L
    monkeypatch.setenv("PM_LSAP_COLUMN_REPAIR", mode)
    monkeypatch.setattr(L, "CORE_EDGES_PER_ROW", 4)                   # a starved core: pricing rounds are certain
    taken = {"plain": 0, "cheap columns": 0}
    for seed in range(24):
        rng = np.random.default_rng(1000 + seed)
        nr = int(rng.integers(150, 900))
        nc = nr if seed % 2 == 0 else nr + int(rng.integers(1, 300))
        U = rng.random((nr, nc))
        kind = "cheap columns" if seed % 4 >= 2 else "plain"
        if kind == "cheap columns":
            U[:, rng.choice(nc, 5, replace=False)] *= 0.05
        info = {}
        sol = L.solve_core(HostMatrix(U), info)
        assert sol is not None and L.certify(HostMatrix(U), *sol, info=info), (seed, info)
        assert np.array_equal(sol[2], scipy_lsa(U)[1]), seed
        taken[kind] += info["column_repairs"]
    if mode == "0":
        assert taken == {"plain": 0, "cheap columns": 0}
    elif mode == "force":
        assert taken["plain"] > 0 and taken["cheap columns"] > 0
    else:
        assert taken["cheap columns"] > 0, taken                     # (by its criterion: many violated rows on few real-held columns)


def test_listed_certificate_grants_a_matched_entry_no_more_than_the_stated_bound():
    """certify_listed's premise |R - C| <= cost_delta is checked on the matched entries (u' - u).  The proof that an UNLISTED entry
    stays further than eps_collect from tight needs |u' - u| <= cost_delta + delta — not the 2 cost_delta + delta round 4 granted: a
    matched entry that is off by 1.5 cost_delta is refused (cost_delta_exceeded), one off by 0.9 cost_delta passes."""
    from platymatch_amd import lsap as L
    rng = np.random.default_rng(77)
    n = 200
    C = rng.random((n, n)) * rng.random((1, n)) + 0.3 * rng.random((n, 1)) + 0.05 * rng.random((n, n))
    cost_delta = 1e-9
    M = HostMatrix(C)                                             # solved on the exact matrix: u is tight on C's matched entries
    sol = L.solve_core(M)
    c4r = sol[2]
    for off, expect in ((0.9, True), (1.5, False)):
        def entries(rows, cols):
            out = C[rows, cols].copy()
            hit = (np.asarray(rows) == 7) & (np.asarray(cols) == c4r[7])
            out[hit] += off * cost_delta                          # the "exact" matrix differs from the solved one at ONE matched entry
            return (out,)
        info = {}
        ok = L.certify_listed(M, *sol, exact_entries=entries, cost_delta=cost_delta, infos=[info])[0]
        assert ok is expect or ok == expect, (off, info)
        assert ("cost_delta_exceeded" in info) == (not expect), (off, info)
