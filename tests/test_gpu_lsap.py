"""The device-resident assignment solve (platymatch_amd/lsap.py) on the GPU: the two kernels that read the dense matrix
against their NumPy restatement, and solve_on_device / solve_eight_on_device against scipy.optimize.linear_sum_assignment —
the call the widget makes (_dock_widget.py:604-611) — on random, rectangular, tied and reference matrices."""
import numpy as np
import pytest
from scipy.optimize import linear_sum_assignment as scipy_lsa

from conftest import load_golden, synth_pair
from test_lsap_core import HostMatrix

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import torch
    from platymatch_amd import _native as nat
    from platymatch_amd.build import build_native
    build_native()
    nat.load()
    assert torch.cuda.is_available()
    return lambda x, dtype=None: nat.to_dev(x, dtype=dtype or torch.float64)


@pytest.mark.parametrize("shape", [(7, 5), (300, 300), (257, 1000), (1000, 257), (1500, 1600)])
def test_kernels_equal_their_numpy_restatement(dev, shape):
    from platymatch_amd import lsap as L
    rng = np.random.default_rng(sum(shape))
    U = rng.random(shape)
    n, m = shape
    W = U if n <= m else np.ascontiguousarray(U.T)
    D, H = L.DeviceMatrix(dev(W)), HostMatrix(W)
    v = -rng.random(W.shape[1]) * 0.1
    for vv, k in ((None, 48), (v, 8), (v, 256)):
        dc, dcost, dflag = D.row_select(vv, k)
        hc, hcost, hflag = H.row_select(vv, k)
        assert dflag == hflag == 0 and np.array_equal(dc, hc) and np.array_equal(dcost, hcost)
    rows = rng.choice(W.shape[0], min(W.shape[0], 200), replace=False).astype(np.int32)
    dj, du1, du2 = D.bid(v, rows)
    hj, hu1, hu2 = H.bid(v, rows)
    assert np.array_equal(dj, hj) and np.array_equal(du1, hu1) and np.array_equal(du2, hu2)
    Wt = W.copy()
    Wt[:, 1 % W.shape[1]] = Wt[:, 0]                     # two columns tie everywhere: lowest column wins, second = first
    tj, tu1, tu2 = L.DeviceMatrix(dev(Wt)).bid(np.zeros(W.shape[1]), rows)
    hj, hu1, hu2 = HostMatrix(Wt).bid(np.zeros(W.shape[1]), rows)
    assert np.array_equal(tj, hj) and np.array_equal(tu1, hu1) and np.array_equal(tu2, hu2)
    assert np.array_equal(D.entries(rows, dj), W[rows, dj])
    sol = L.solve_core(D)
    assert sol is not None
    u, v2, c4r = sol
    scale = max(abs(u).max(), abs(v2).max())
    for du, dv, dc4 in ((u, v2, c4r), (u + 1e-6, v2, c4r), (u, v2, np.roll(c4r, 1))):
        cap = 8 * W.shape[1] + 1024
        d = D.certificate(du, dv, dc4, 1e-13 * scale, 1e-7 * scale, cap)
        h = H.certificate(du, dv, dc4, 1e-13 * scale, 1e-7 * scale, cap)
        assert d[0] == h[0] and d[1] == h[1]
        if h[2] is not None:
            assert sorted(zip(map(tuple, d[2]), d[3])) == sorted(zip(map(tuple, h[2]), h[3]))
        assert abs(d[4] - h[4]) <= 1e-12 * max(abs(h[4]), 1e-300) or (np.isnan(d[4]) and np.isnan(h[4]))
    Wn = W.copy()
    Wn[W.shape[0] // 2, 3] = np.nan
    assert L.DeviceMatrix(dev(Wn)).row_select(None, 8)[2] == 1


@pytest.mark.parametrize("shape", [(40, 40), (64, 900), (700, 90), (1200, 1200), (1100, 1300), (1, 5), (5, 1)])
def test_solve_on_device_equals_scipy(dev, shape):
    from platymatch_amd import lsap as L
    rng = np.random.default_rng(sum(shape) + 1)
    for trial in range(3):
        U = rng.random(shape) if trial else rng.random(shape) * rng.random((1, shape[1])) + 0.3 * rng.random((shape[0], 1))
        info = {}
        r, c = L.solve_on_device(dev(U), info=info, force=True)
        rs, cs = scipy_lsa(U)
        assert np.array_equal(r, rs) and np.array_equal(c, cs), info
        assert info["route"] == "device", info
    # a strided view (row block of a larger allocation) is accepted as it is
    big = dev(rng.random((shape[0], shape[1] + 5)))
    r, c = L.solve_on_device(big[:, :shape[1]], force=True)
    rs, cs = scipy_lsa(big[:, :shape[1]].cpu().numpy())
    assert np.array_equal(r, rs) and np.array_equal(c, cs)


def test_tied_and_invalid_matrices_take_scipys_own_algorithm(dev):
    from platymatch_amd import lsap as L
    rng = np.random.default_rng(0)
    for trial in range(25):
        n, m = int(rng.integers(2, 60)), int(rng.integers(2, 60))
        U = rng.integers(0, 4, size=(n, m)).astype(np.float64)
        info = {}
        r, c = L.solve_on_device(dev(U), info=info, force=True)
        rs, cs = scipy_lsa(U)
        assert np.array_equal(r, rs) and np.array_equal(c, cs), (trial, info)         # identical indices, ties included
    U = rng.random((40, 40))
    U[7] = U[3]                                                                       # duplicate nuclei: identical rows
    info = {}
    r, c = L.solve_on_device(dev(U), info=info, force=True)
    assert info["route"] == "host" and np.array_equal(c, scipy_lsa(U)[1])
    U[2, 2] = np.nan
    with pytest.raises(ValueError):
        L.solve_on_device(dev(U), force=True)
    U[2, 2] = np.inf                                                                  # forbidden edge: SciPy accepts it
    r, c = L.solve_on_device(dev(U), info=info, force=True)
    assert info["route"] == "host" and np.array_equal(c, scipy_lsa(U)[1])


@pytest.mark.parametrize("name", ["insitu02_affine", "insitu04_affine", "synth96x128", "synth128"])
def test_reference_scenarios_through_the_device_route(dev, name):
    """Every hypothesis of a reference scenario, forced through the device route: the assignment vectors the reference's
    SciPy calls produced."""
    from platymatch_amd import lsap as L, pipeline as P
    d = load_golden(name)
    be = P.GpuBackend()
    U, _ = P.build_costs(be, be.cloud(d["moving"]), be.cloud(d["fixed"]))
    for h in range(8):
        info = {}
        r, c = L.solve_on_device(U[h], info=info, force=True)
        assert info["route"] == "device", (h, info)
        assert np.array_equal(r, d["lsa_rows"][h]) and np.array_equal(c, d["lsa_cols"][h]), h


def test_eight_assignments_of_a_3000_point_pair_equal_scipy(dev):
    from platymatch_amd import lsap as L, pipeline as P
    mv, fx, _ = synth_pair(3000, 77)
    be = P.GpuBackend()
    U, _ = P.build_costs(be, be.cloud(mv), be.cloud(fx[:, :2900]))
    info = {}
    got = L.solve_eight_on_device(U, info=info)
    assert all(r.startswith("device") for r in info["routes"]), info["routes"]
    assert sum("sibling" in r for r in info["routes"]) == 4                            # four solves served eight hypotheses
    for h in range(8):
        rs, cs = scipy_lsa(U[h].cpu().numpy())
        assert np.array_equal(got[h][0], rs) and np.array_equal(got[h][1], cs), h
    print({k: info["details"][0].get(k) for k in ("rounds", "edges", "steps", "violated_per_round", "max_matched_slack", "tight")})


def test_config2_all_eight_assignments_of_a_5000_point_pair_equal_scipy(dev):
    """BASELINE config 2 literally ("5k-nucleus pair: shape-context + chi-square cost + Hungarian on 1 MI355X"): all eight
    assignments of a 5 000 x 5 000 synthetic pair by the device-resident route == scipy.optimize.linear_sum_assignment on the
    same device-built matrices copied to the host (_dock_widget.py:604-611) — the dense solve the certificate stands in for."""
    from platymatch_amd import lsap as L, pipeline as P
    mv, fx, _ = synth_pair(5000, 42)
    be = P.GpuBackend()
    U, _ = P.build_costs(be, be.cloud(mv), be.cloud(fx))
    info = {}
    got = L.solve_eight_on_device(U, info=info)
    assert all(r.startswith("device") for r in info["routes"]), info["routes"]
    for h in range(8):
        rs, cs = scipy_lsa(U[h].cpu().numpy())
        assert np.array_equal(got[h][0], rs) and np.array_equal(got[h][1], cs), h


def test_eight_assignments_with_more_moving_than_fixed_points(dev):
    """N > M: the solver works on the transposed matrices (rows = the short side, as SciPy does); answers in U's own indexing."""
    from platymatch_amd import lsap as L, pipeline as P
    mv, fx, _ = synth_pair(1400, 78)
    be = P.GpuBackend()
    U, _ = P.build_costs(be, be.cloud(mv), be.cloud(fx[:, :1150]))
    info = {}
    got = L.solve_eight_on_device(U, info=info)
    assert all(r.startswith("device") for r in info["routes"]), info["routes"]
    for h in range(8):
        rs, cs = scipy_lsa(U[h].cpu().numpy())
        assert len(got[h][0]) == 1150 and np.array_equal(got[h][0], rs) and np.array_equal(got[h][1], cs), h


def test_fuzz_of_distributions_and_shapes_against_scipy(dev):
    """Negative costs, large offsets, tiny scales, heavy tails, low-rank structure, near-ties, wide and tall shapes: whatever
    route is taken, the indices are SciPy's."""
    from platymatch_amd import lsap as L
    rng = np.random.default_rng(11)
    routes = {"device": 0, "host": 0}
    for trial in range(120):
        n, m = int(rng.integers(1, 400)), int(rng.integers(1, 400))
        kind = trial % 8
        if kind == 0:
            U = rng.normal(size=(n, m))
        elif kind == 1:
            U = rng.random((n, m)) + 1e6
        elif kind == 2:
            U = rng.random((n, m)) * 1e-200
        elif kind == 3:
            U = rng.standard_cauchy(size=(n, m))
        elif kind == 4:
            U = np.outer(rng.random(n), rng.random(m)) + 1e-3 * rng.random((n, m))
        elif kind == 5:
            U = np.round(rng.random((n, m)), 2)                      # many ties
        elif kind == 6:
            U = np.abs(rng.normal(size=(n, 1)) - rng.normal(size=(1, m)))     # 1-D geometry: structured, near-ties
        else:
            U = -rng.random((n, m)) ** 3
        info = {}
        r, c = L.solve_on_device(dev(U), info=info, force=True)
        rs, cs = scipy_lsa(U)
        assert np.array_equal(r, rs) and np.array_equal(c, cs), (trial, kind, n, m, info)
        routes[info["route"]] += 1
    assert routes["device"] >= 60, routes                              # generic matrices are certified; ties go to the host


def test_pairing_cost_kernel_equals_the_eight_matrix_launch(dev):
    """One hypothesis and its twin at a time (pm_chi2_cost_pair_sym / the general kernel twice) against the rows of the
    eight-matrix build, bit for bit, on generic descriptors and on a set that breaks the frame-permutation relation."""
    import torch
    from platymatch_amd import _kernels as K, pipeline as P
    mv, fx, _ = synth_pair(700, 31)
    be = P.GpuBackend()
    sc_m, sc_f, _ = P.build_descriptors(be, be.cloud(mv), be.cloud(fx[:, :650]))
    U8 = K.chi2_cost8(sc_m, sc_f)
    assert K.chi2_symmetric(sc_m, sc_f)
    for t, (h, twin) in enumerate(K.PAIRINGS):
        for sym in (True, False):
            U2 = K.chi2_cost_pair(sc_m, sc_f, t, sym)
            assert torch.equal(U2[0], U8[h]) and torch.equal(U2[1], U8[twin]), (t, sym)
        U2f = K.chi2_cost_pair(sc_m[:1], sc_f[:1], t, True)                    # frame 1 only is enough for the half-cost path
        assert torch.equal(U2f[0], U8[h]) and torch.equal(U2f[1], U8[twin])
    broken = sc_f.clone()
    broken[2, 5, 7] += 0.25                                                     # frame 3 no longer a permutation of frame 1
    assert not K.chi2_symmetric(sc_m, broken)
    U8b = K.chi2_cost8(sc_m, broken)
    for t, (h, twin) in enumerate(K.PAIRINGS):
        U2 = be.chi2_cost_pair(sc_m, broken, t)
        assert torch.equal(U2[0], U8b[h]) and torch.equal(U2[1], U8b[twin]), t


def test_streamed_hypotheses_give_the_same_registration(dev):
    """estimate_transform with two cost matrices resident at a time (the mode for clouds whose eight matrices exceed HBM)
    against the default: assignment vectors, inlier counts and 4x4 matrices identical."""
    from platymatch_amd import pipeline as P
    from platymatch_amd.estimate_transform import perform_icp as pi
    pi.VERBOSE = False
    mv, fx, _ = synth_pair(1700, 33)
    kw = dict(ransac_trials=400, ransac_error=16, icp_iterations=10, seed=2)
    for fixed in (fx[:, :1650], fx):                                            # N > M and N == M
        d0, d1 = {}, {}
        a = P.estimate_transform(mv, fixed, details=d0, cost_mode='exact', options={"stream_hypotheses": False}, **kw)
        b = P.estimate_transform(mv, fixed, details=d1, options={"stream_hypotheses": True}, **kw)        # (the default cost mode)
        assert d1["assignment"]["mode"].startswith("streamed")
        for h in range(8):
            assert np.array_equal(d0["lsa"][h][0], d1["lsa"][h][0]) and np.array_equal(d0["lsa"][h][1], d1["lsa"][h][1]), h
        assert np.array_equal(a[2], b[2]) and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_near_tie_beyond_the_dense_solvers_reach_is_settled_on_its_block(dev, monkeypatch):
    """Where SciPy's dense algorithm is no fallback for the whole matrix (lsap.DENSE_FALLBACK_MAX_ENTRIES; lowered here), a
    hypothesis whose optimum is certified but has an alternative inside the margin is SETTLED (round 4): the rows its near-tight
    entries connect are assigned by SciPy's algorithm on their own block and spliced in — the reference never refuses
    (_dock_widget.py:604-611).  The engineered matrix has one 2-cycle worth 1e-14; the answer equals SciPy's on the whole matrix.
    Only when the block cannot be formed (here: RESOLVE_MAX_BLOCK_ROWS lowered) does the hypothesis come back None (the driver
    raises) — or, with accept_near_ties=True, as the certified optimum, labelled as such."""
    from platymatch_amd import lsap as L
    rng = np.random.default_rng(21)
    n = 1100
    base = rng.random((n, n)) + 0.5
    u, v, c = L.solve_core(L.DeviceMatrix(dev(base)))
    rs, cs = None, None
    for sign in (+1.0, -1.0):                        # the alternative 1e-14 dearer than the start's optimum, then 1e-14 cheaper
        U = base.copy()
        i1, i2 = 5, 900
        U[i1, c[i2]] = (u[i1] + v[c[i2]]) + sign * 0.5e-14
        U[i2, c[i1]] = (u[i2] + v[c[i1]]) + sign * 0.5e-14
        Ud = dev(U)
        rs, cs = scipy_lsa(U)
        monkeypatch.setattr(L, "DENSE_FALLBACK_MAX_ENTRIES", 0)
        ih, it = {}, {}
        out = L.solve_pair_on_device(Ud, Ud, ih, it)
        assert ih["optimal"] and "settled" in ih["route"] and ih["resolved_groups"] == [2], ih.get("route")
        for r_, c_ in out:
            assert np.array_equal(r_, rs) and np.array_equal(c_, cs), sign            # SciPy's own pick between the two
        monkeypatch.undo()
    # the block cannot be formed: refused by default, the certified optimum with accept_near_ties
    monkeypatch.setattr(L, "DENSE_FALLBACK_MAX_ENTRIES", 0)
    monkeypatch.setattr(L, "RESOLVE_MAX_BLOCK_ROWS", 1)
    ih, it = {}, {}
    out = L.solve_pair_on_device(Ud, Ud, ih, it)
    assert out == [None, None] and ih["optimal"] and ih["route"].startswith("uncertified (too large")
    ih, it = {}, {}
    out = L.solve_pair_on_device(Ud, Ud, ih, it, accept_near_ties=True)
    assert "near-tie" in ih["route"] and "near-tie" in it["route"]
    for r_, c_ in out:
        assert np.array_equal(r_, rs) and sorted(c_) == list(range(n))
        assert abs(U[r_, c_].sum() - U[rs, cs].sum()) <= 1e-12 * n                   # optimal; which of the two near-equal ones is open
    # the driver's eight-matrix path raises rather than returning a hole
    from platymatch_amd import pipeline as P
    import torch
    U8 = torch.stack([Ud] * 8)
    with pytest.raises(RuntimeError, match="accept_near_ties"):
        P.assign(U8, [0, n])
    lsa = P.assign(U8, [0, n], accept_near_ties=True)
    assert all(a is not None for a in lsa)
    monkeypatch.setattr(L, "RESOLVE_MAX_BLOCK_ROWS", 4096)
    lsa = P.assign(U8, [0, n])                                                        # settled: no flag needed
    assert all(np.array_equal(a[1], cs) for a in lsa)
    # with the dense solver allowed again the same matrix takes SciPy's own algorithm: identical indices
    monkeypatch.undo()
    ih = {}
    out = L.solve_pair_on_device(Ud, Ud, ih, {})
    assert ih["route"] == "host" and np.array_equal(out[0][1], cs)


def test_cost_build_by_pairings_writes_the_same_eight_matrices(dev):
    """The pipelined build (one launch per pairing into the eight-matrix buffer, an event each) against the single launch,
    and the assignment started from the events against the one started after a full synchronise."""
    import torch
    from platymatch_amd import _kernels as K, lsap as L
    mv, fx, _ = synth_pair(1500, 5)
    x, y = dev(mv), dev(np.ascontiguousarray(fx[:, :1400]))
    hm = K.shape_context(x, K.centroid(x), K.pca_axis(x), K.mean_distance(x), 2)["hist"]
    hf = K.shape_context(y, K.centroid(y), K.pca_axis(y), K.mean_distance(y), 4)["hist"]
    assert K.chi2_symmetric(hm, hf)
    U = K.chi2_cost8_frame1(hm[0], hf[0])
    U2, events = K.chi2_cost8_frame1_by_pairings(hm[0], hf[0])
    a = L.solve_eight_on_device(U2, ready=events)                 # starts while later pairings may still be in flight
    torch.cuda.synchronize()
    assert len(events) == 4 and torch.equal(U, U2)
    b = L.solve_eight_on_device(U)
    for h in range(8):
        assert np.array_equal(a[h][0], b[h][0]) and np.array_equal(a[h][1], b[h][1])
        assert np.array_equal(a[h][1], scipy_lsa(U[h].cpu().numpy())[1])


@pytest.mark.parametrize("shape", [(1, 1), (63, 65), (64, 64), (300, 1000), (1000, 300), (2049, 777)])
def test_tiled_transpose(dev, shape):
    import torch
    from platymatch_amd import lsap as L
    U = dev(np.random.default_rng(sum(shape)).random(shape))
    assert torch.equal(L.transposed(U), U.t().contiguous())
    V = dev(np.random.default_rng(1).random((shape[0], shape[1] + 5)))[:, :shape[1]]      # a view with a row pitch
    assert torch.equal(L.transposed(V), V.t().contiguous())


@pytest.mark.parametrize("n,m", [(4200, 4000), (4000, 4200), (4100, 4060)])
def test_rectangular_chi_square_assignments_at_a_few_thousand_nuclei(dev, n, m):
    """Clouds of different sizes (the rule for real specimen pairs), both orientations and a near-square one, at a size where the
    auction warm start, its reverse steps, the column-side searches for stranded columns and (N > M) the tiled transpose all do
    real work: two hypotheses and their twins against SciPy, routes device-resident, counters sane."""
    from platymatch_amd import lsap as L, pipeline as P
    mv, fx, _ = synth_pair(max(n, m), 31)
    be = P.GpuBackend()
    U, _ = P.build_costs(be, be.cloud(np.ascontiguousarray(mv[:, :n])), be.cloud(np.ascontiguousarray(fx[:, :m])))
    for h, twin in ((0, 5), (2, 7)):
        ih, it = {}, {}
        got = L.solve_pair_on_device(U[h], U[twin], ih, it)
        assert ih["route"] == "device" and it["route"].startswith("device"), (ih.get("route"), it.get("route"))
        assert "auction_bids" in ih and ih["dummy_scans"] % 1000000 <= 4 * abs(n - m) + 16, ih
        for k, hyp in enumerate((h, twin)):
            rs, cs = scipy_lsa(U[hyp].cpu().numpy())
            assert np.array_equal(got[k][0], rs) and np.array_equal(got[k][1], cs), (hyp, ih)


@pytest.mark.parametrize("shape", [(1200, 1200), (1100, 1300), (2500, 2500)])
def test_native_driver_gives_the_python_drivers_answers(dev, shape, monkeypatch):
    """csrc/pm_lsap_resident.hip (round 4): solve_core + certify as one foreign call each.  Same kernels, same core solver, same
    sequence: the certified assignment is the Python driver's (and SciPy's), the duals are feasible to the same tolerance, the
    certificate's verdict and counters agree."""
    from platymatch_amd import lsap as L
    rng = np.random.default_rng(shape[0] + 3)
    U = rng.random(shape) * rng.random((1, shape[1])) + 0.3 * rng.random((shape[0], 1))
    Ud = dev(U)
    res = {}
    for native in (False, True):
        monkeypatch.setattr(L, "NATIVE_DRIVER", native)
        W = L.DeviceMatrix(Ud)
        info = {}
        sol = L.solve_core(W, info)
        assert sol is not None and (info.get("driver") == "native") == native
        ok = L.certify(W, *sol, info=info)
        res[native] = (sol, ok, info)
    (sp, okp, ip), (sn, okn, inn) = res[False], res[True]
    assert okp and okn and np.array_equal(sp[2], sn[2]) and np.array_equal(sn[2], scipy_lsa(U)[1])
    assert ip["violations"] == inn["violations"] == 0 and ip["loose"] == inn["loose"] == 0
    assert inn["optimal"] and inn["unique"] and inn["tight_within_eps"] >= 0
    # the native certificate rejects what the Python one rejects: shifted duals, a rotated matching
    monkeypatch.setattr(L, "NATIVE_DRIVER", True)
    W = L.DeviceMatrix(Ud)
    u, v, c = sn
    assert not L.certify(W, u + 1e-6, v, c) and not L.certify(W, u, v, np.roll(c, 1))
    # non-finite entries: no solve (the caller takes SciPy's own path)
    Un = U.copy()
    Un[3, 4] = np.nan
    assert L.solve_core(L.DeviceMatrix(dev(Un))) is None


def test_warm_up_runs_the_selection_code_once_per_device_and_leaves_no_trace(dev):
    """platymatch_amd.warm_up / lsap.warm_up: the first-use costs of the selection code (torch's own kernels behind it) paid on a toy
    matrix ahead of the first registration — idempotent, touches no random generator, and a real solve afterwards is SciPy's."""
    import torch
    import platymatch_amd
    from platymatch_amd import lsap as L
    state = np.random.get_state()[1].copy()
    gen = torch.cuda.get_rng_state().clone()
    platymatch_amd.warm_up()
    platymatch_amd.warm_up()                                          # a no-op ever after
    assert torch.cuda.current_device() in L._WARM["done"] and not L._WARM["threads"]
    assert np.array_equal(np.random.get_state()[1], state) and torch.equal(torch.cuda.get_rng_state(), gen)
    L.warm_up("cpu")                                                  # nothing to warm: returns
    rng = np.random.default_rng(11)
    U = rng.random((1100, 1300))
    r, c = L.solve_on_device(dev(U))
    assert np.array_equal(c, scipy_lsa(U)[1])
