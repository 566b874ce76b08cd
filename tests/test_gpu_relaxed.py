"""The builds cost_mode='auto' (THE DEFAULT since round 5) starts from instead of the eight exact matrices: the relaxed-rounding
float64 build (pm_chi2_cost8_relaxed; 1 024..8 191 nuclei) and the packed-float32 filter (pm_chi2_filter4; from 8 192).  Statements:
every relaxed / filter entry lies within the stated bound of the exact one (the exact one being the reference's bits,
shape_context.py:88-99); the twins coincide; and a registration through either returns the SAME assignment vectors, inlier counts
and A_sc as cost_mode='exact' — by proof: the assignment is certified against the EXACT matrix on its matched and near-tight
entries (pm_chi2_entries_sym, lsap.certify_listed: the exact mode's own margins); a pairing that does not certify is built exactly."""
import numpy as np
import pytest

from conftest import synth_pair

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    import torch
    from platymatch_amd import _kernels as K, _native as nat, lsap as L, pipeline as P
    from platymatch_amd.estimate_transform import perform_icp as pi
    nat.load()
    assert torch.cuda.is_available()
    pi.VERBOSE = False

    class G:
        pass
    G.K, G.nat, G.L, G.P, G.t = K, nat, L, P, torch
    return G


@pytest.mark.parametrize("n,m", [(3000, 3000), (1500, 2100), (2100, 1500), (70, 90)])
def test_relaxed_entries_are_within_the_bound_and_twins_coincide(g, n, m):
    mv, fx, _ = synth_pair(max(n, m), 5 + n)
    be = g.P.GpuBackend()
    sc_m, sc_f, _ = g.P.build_descriptors(be, be.cloud(np.ascontiguousarray(mv[:, :n])), be.cloud(np.ascontiguousarray(fx[:, :m])))
    assert g.K.chi2_symmetric(sc_m, sc_f)
    exact = g.K.chi2_cost8(sc_m, sc_f)
    delta = g.K.chi2_relaxed_delta()
    for variant in (0, 1, 2):
        rel = g.K.chi2_cost8_relaxed(sc_m[0], sc_f[0], variant=variant)
        assert float((rel - exact).abs().max()) <= delta, variant
        for twin, h in g.L.TWINS.items():
            assert g.t.equal(rel[h], rel[twin]), (variant, h)
    # a NaN descriptor row (a point on the centroid) is NaN in both builds
    bad = sc_m[0].clone()
    bad[3] = float("nan")
    rel = g.K.chi2_cost8_relaxed(bad, sc_f[0], variant=1)
    assert bool(g.t.isnan(rel[:, 3]).all()) and bool(g.t.isfinite(rel[:, 4]).all())


@pytest.mark.parametrize("n,m,seed", [(3000, 2900, 77), (5000, 5000, 42), (2400, 2600, 3), (1200, 1200, 9), (600, 650, 1)])
def test_relaxed_mode_returns_the_exact_modes_registration(g, n, m, seed):
    mv, fx, _ = synth_pair(max(n, m), seed)
    mv, fx = np.ascontiguousarray(mv[:, :n]), np.ascontiguousarray(fx[:, :m])
    kw = dict(ransac_trials=400, icp_iterations=6, seed=5)
    de, dr = {}, {}
    a = g.P.estimate_transform(mv, fx, details=de, cost_mode='exact', **kw)
    b = g.P.estimate_transform(mv, fx, details=dr, cost_mode='relaxed', **kw)
    for h in range(8):
        assert np.array_equal(de["lsa"][h][0], dr["lsa"][h][0]) and np.array_equal(de["lsa"][h][1], dr["lsa"][h][1]), h
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    if min(n, m) >= g.P.RELAXED_MIN_POINTS:
        modes = [d.get("cost_mode", "") for d in dr["assignment"]["details"]]
        assert all(x.startswith("relaxed") or x.startswith("exact (rebuilt") for x in modes), modes
        print("%d x %d: %d of 8 hypotheses certified on the relaxed matrices, %d rebuilt exactly" % (n, m, sum(x.startswith("relaxed") for x in modes),
                                                                                                   sum(x.startswith("exact") for x in modes)))
        # generic clouds have no near-ties: every hypothesis is settled without an exact matrix being built
        assert all("listed entries" in x for x in modes), modes
    else:
        assert "details" not in dr["assignment"] or all("cost_mode" not in d for d in dr["assignment"]["details"])


def test_a_pairing_that_does_not_certify_is_rebuilt_exactly(g, monkeypatch):
    """An absurd error bound (the margin can never be shown) sends every pairing to the exact rebuild: identical results, and the
    buffer then holds the exact matrices' bits."""
    mv, fx, _ = synth_pair(2000, 11)
    be = g.P.GpuBackend()
    sc_m, sc_f, _ = g.P.build_descriptors(be, be.cloud(mv), be.cloud(fx))
    exact = g.K.chi2_cost8(sc_m, sc_f)
    want = g.L.solve_eight_on_device(exact.clone())
    U = g.K.chi2_cost8_relaxed(sc_m[0], sc_f[0], variant=2)
    info = {}
    got = g.L.solve_eight_on_device(U, info=info, min_eps=1.0,
                                    exact_rebuild=lambda h: g.K.chi2_cost_pair_into(sc_m[0], sc_f[0], [p[0] for p in g.K.PAIRINGS].index(h), U))
    assert all(d["cost_mode"].startswith("exact (rebuilt") for d in info["details"])
    assert g.t.equal(U, exact)
    for h in range(8):
        assert np.array_equal(got[h][1], want[h][1]), h


@pytest.mark.parametrize("n,m", [(1500, 2100), (2100, 1500), (700, 700)])
def test_listed_entries_carry_the_exact_matrices_bits(g, n, m):
    """pm_chi2_entries_sym: random (row, col) lists against the entries of the exact eight-matrix build, natural and rolled order
    of every pairing, bit for bit; out-of-range indices are refused by the wrapper."""
    mv, fx, _ = synth_pair(max(n, m), 13 + n)
    be = g.P.GpuBackend()
    sc_m, sc_f, _ = g.P.build_descriptors(be, be.cloud(np.ascontiguousarray(mv[:, :n])), be.cloud(np.ascontiguousarray(fx[:, :m])))
    assert g.K.chi2_symmetric(sc_m, sc_f)
    exact = g.K.chi2_cost8(sc_m, sc_f).cpu().numpy()
    rng = np.random.default_rng(n + m)
    rows = np.concatenate([rng.integers(0, n, 5000), [0, n - 1, 0, n - 1]])
    cols = np.concatenate([rng.integers(0, m, 5000), [0, m - 1, m - 1, 0]])
    for t, (h, twin) in enumerate(g.K.PAIRINGS):
        nat_v, rol_v = (x.cpu().numpy() for x in g.K.chi2_entries(sc_m[0], sc_f[0], t, rows, cols))
        assert np.array_equal(nat_v.view(np.uint64), exact[h][rows, cols].view(np.uint64)), t
        assert np.array_equal(rol_v.view(np.uint64), exact[twin][rows, cols].view(np.uint64)), t
    with pytest.raises(ValueError):
        g.K.chi2_entries(sc_m[0], sc_f[0], 0, np.array([n]), np.array([0]))
    empty = g.K.chi2_entries(sc_m[0], sc_f[0], 0, np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64))
    assert empty[0].numel() == 0


def test_listed_certificate_at_a_size_where_the_wide_margin_fails(g):
    """20 000 nuclei: the first form of the relaxed mode (a uniqueness margin of 2 N delta on the relaxed matrix) and the listed
    certificate, side by side on the same relaxed build — the listed one settles all eight hypotheses with the exact mode's
    answers; how many the wide margin settles is printed (at 50 000: about half, profiles/r04_chi2_relaxed.txt)."""
    n = 20000
    mv, fx, _ = synth_pair(n, 42)
    be = g.P.GpuBackend()
    sc_m, sc_f, _ = g.P.build_descriptors(be, be.cloud(mv), be.cloud(fx))
    exact = g.K.chi2_cost8(sc_m, sc_f)
    want = g.L.solve_eight_on_device(exact)
    del exact
    U = g.K.chi2_cost8_relaxed(sc_m[0], sc_f[0], variant=2)
    delta = g.K.chi2_relaxed_delta()
    pairing_of = {p[0]: t for t, p in enumerate(g.K.PAIRINGS)}
    asked = []

    def entries(h):
        def fetch(rows, cols):
            asked.append(len(rows))
            return tuple(x.cpu().numpy() for x in g.K.chi2_entries(sc_m[0], sc_f[0], pairing_of[h], rows, cols))
        return fetch

    def no_rebuild(h):
        raise AssertionError("pairing %d was sent to the exact rebuild" % h)
    info = {}
    got = g.L.solve_eight_on_device(U, info=info, exact_entries=entries, cost_delta=delta, exact_rebuild=no_rebuild)
    for h in range(8):
        assert np.array_equal(got[h][0], want[h][0]) and np.array_equal(got[h][1], want[h][1]), h
    assert all("listed entries" in d["cost_mode"] for d in info["details"])
    assert max(asked) < 12 * n                                     # a few entries per row were evaluated exactly, not 4e8
    info2 = {}
    g.L.solve_eight_on_device(U, info=info2, allow_host=False, min_eps=2.0 * n * delta, exact_rebuild=lambda h: None)
    print("listed certificate: 8 of 8; 2 N delta margin: %d of 8 (margin %.1e); exact entries evaluated per pairing: %s"
          % (sum(d["cost_mode"].startswith("relaxed") for d in info2["details"]), 2.0 * n * delta, asked))


def test_tied_clouds_fall_back_to_the_exact_build_and_give_the_exact_modes_answer(g):
    """Duplicate nuclei make whole cost rows equal: every hypothesis has tied optima, nothing can be certified unique on the relaxed
    build, so each pairing is rebuilt exactly and goes the exact mode's way (SciPy's algorithm on the host settles the ties) —
    the two modes must return the same assignment vectors, inlier counts and transforms."""
    n = 1400
    mv, fx, _ = synth_pair(n, 23)
    mv, fx = mv.copy(), fx.copy()
    mv[:, 100:112] = mv[:, 200:212]                     # twelve duplicated nuclei in the moving cloud
    fx[:, 300:306] = fx[:, 900:906]                     # six in the fixed one
    kw = dict(ransac_trials=300, icp_iterations=4, seed=3)
    de, dr = {}, {}
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", g.P.EdgeGuardWarning)
        a = g.P.estimate_transform(mv, fx, details=de, cost_mode='exact', **kw)
        b = g.P.estimate_transform(mv, fx, details=dr, cost_mode='relaxed', **kw)
    for h in range(8):
        assert np.array_equal(de["lsa"][h][0], dr["lsa"][h][0]) and np.array_equal(de["lsa"][h][1], dr["lsa"][h][1]), h
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    modes = [d.get("cost_mode", "") for d in dr["assignment"]["details"]]
    assert all(x.startswith("exact (rebuilt") for x in modes), modes


@pytest.mark.parametrize("n,m", [(1500, 2100), (2100, 1500), (3000, 3000)])
def test_filter_build_is_within_its_bound_of_both_orders_of_every_pairing(g, n, m):
    """pm_chi2_filter4 (packed float32 terms): matrix t against the exact natural-order matrix AND its rolled-order twin."""
    mv, fx, _ = synth_pair(max(n, m), 31 + n)
    be = g.P.GpuBackend()
    sc_m, sc_f, _ = g.P.build_descriptors(be, be.cloud(np.ascontiguousarray(mv[:, :n])), be.cloud(np.ascontiguousarray(fx[:, :m])))
    assert g.K.chi2_symmetric(sc_m, sc_f)
    exact = g.K.chi2_cost8(sc_m, sc_f)
    F = g.K.chi2_filter4(sc_m[0], sc_f[0])
    delta = g.K.chi2_filter_delta()
    worst = 0.0
    for t, (h, twin) in enumerate(g.K.PAIRINGS):
        for k in (h, twin):
            worst = max(worst, float((F[t] - exact[k]).abs().max()))
    print("%d x %d: largest |filter - exact| = %.2e (bound %.1e)" % (n, m, worst, delta))
    assert worst <= delta
    # float32 storage (the product's): within the bound too, and the pairing-by-pairing launches give the four-matrix launch's values
    F32 = g.K.chi2_filter4(sc_m[0], sc_f[0], dtype=g.t.float32)
    for t, (h, twin) in enumerate(g.K.PAIRINGS):
        for k in (h, twin):
            assert float((F32[t].double() - exact[k]).abs().max()) <= delta
        assert g.t.equal(g.K.chi2_filter_pair(sc_m[0], sc_f[0], t, dtype=g.t.float32), F32[t]), t


@pytest.mark.parametrize("n,m,seed", [(3000, 2900, 77), (5000, 5000, 42), (2400, 2600, 3), (1200, 1200, 9), (600, 650, 1), (9000, 8400, 4)])
def test_filter_mode_returns_the_exact_modes_registration(g, n, m, seed, monkeypatch):
    """estimate_transform(cost_mode='filter'): no exact matrix is built, the assignment is solved with exact costs on the entries a
    float32 build selects and certified against both exact matrices of every pairing — the same assignment vectors, inlier counts
    and transforms as the exact mode, bit for bit.  (The mode hands clouds below FILTER_MIN_POINTS to the relaxed mode; the
    threshold is lowered here so that the small cases run through the filter too; the last case is above it as shipped.)"""
    if min(n, m) < g.P.FILTER_MIN_POINTS:
        monkeypatch.setattr(g.P, "FILTER_MIN_POINTS", g.P.RELAXED_MIN_POINTS)
    mv, fx, _ = synth_pair(max(n, m), seed)
    mv, fx = np.ascontiguousarray(mv[:, :n]), np.ascontiguousarray(fx[:, :m])
    kw = dict(ransac_trials=400, icp_iterations=6, seed=5)
    de, dr = {}, {}
    a = g.P.estimate_transform(mv, fx, details=de, cost_mode='exact', **kw)
    b = g.P.estimate_transform(mv, fx, details=dr, cost_mode='filter', **kw)
    for h in range(8):
        assert np.array_equal(de["lsa"][h][0], dr["lsa"][h][0]) and np.array_equal(de["lsa"][h][1], dr["lsa"][h][1]), h
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    if min(n, m) >= g.P.RELAXED_MIN_POINTS:
        modes = [d.get("cost_mode", "") for d in dr["assignment"]["details"]]
        assert all(x.startswith("filter") for x in modes), modes
        print("%d x %d: %s; polishing rounds %s" % (n, m, modes[0], [d.get("polish_violated") for d in dr["assignment"]["details"][:4]]))


def test_filter_mode_on_tied_clouds_builds_the_pairings_exactly(g, monkeypatch):
    monkeypatch.setattr(g.P, "FILTER_MIN_POINTS", g.P.RELAXED_MIN_POINTS)
    n = 1400
    mv, fx, _ = synth_pair(n, 23)
    mv, fx = mv.copy(), fx.copy()
    mv[:, 100:112] = mv[:, 200:212]
    fx[:, 300:306] = fx[:, 900:906]
    kw = dict(ransac_trials=300, icp_iterations=4, seed=3)
    de, dr = {}, {}
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", g.P.EdgeGuardWarning)
        a = g.P.estimate_transform(mv, fx, details=de, cost_mode='exact', **kw)
        b = g.P.estimate_transform(mv, fx, details=dr, cost_mode='filter', **kw)
    for h in range(8):
        assert np.array_equal(de["lsa"][h][0], dr["lsa"][h][0]) and np.array_equal(de["lsa"][h][1], dr["lsa"][h][1]), h
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    modes = [d.get("cost_mode", "") for d in dr["assignment"]["details"]]
    assert all(x.startswith("exact (built") for x in modes), modes


@pytest.mark.parametrize("n,m,tied", [(9000, 8400, False), (8400, 9000, False), (1500, 1400, True), (1400, 1500, True)])
def test_streamed_filter_mode_builds_one_pairing_at_a_time_and_returns_the_exact_modes_registration(g, n, m, tied, monkeypatch):
    """cost_mode='filter' with the hypotheses streamed (what clouds beyond HBM's four filter matrices get): each pairing's filter
    matrix is built on its own stream when its turn comes — for N > M with the descriptors' roles swapped, which is the transposed
    filter —, solved and certified as in the resident form; pairings that cannot be certified (the tied clouds) are built exactly
    after every filter matrix has been released."""
    if min(n, m) < g.P.FILTER_MIN_POINTS:
        monkeypatch.setattr(g.P, "FILTER_MIN_POINTS", g.P.RELAXED_MIN_POINTS)
    mv, fx, _ = synth_pair(max(n, m), 61 + n)
    mv, fx = np.ascontiguousarray(mv[:, :n]).copy(), np.ascontiguousarray(fx[:, :m]).copy()
    if tied:
        mv[:, 100:108] = mv[:, 200:208]
        fx[:, 300:304] = fx[:, 900:904]
    kw = dict(ransac_trials=300, icp_iterations=4, seed=8)
    de, dr = {}, {}
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", g.P.EdgeGuardWarning)
        a = g.P.estimate_transform(mv, fx, details=de, cost_mode='exact', **kw)
        b = g.P.estimate_transform(mv, fx, details=dr, cost_mode='filter', options={"stream_hypotheses": True}, **kw)
    for h in range(8):
        assert np.array_equal(de["lsa"][h][0], dr["lsa"][h][0]) and np.array_equal(de["lsa"][h][1], dr["lsa"][h][1]), h
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    modes = [d.get("cost_mode", "") for d in dr["assignment"]["details"]]
    assert "streamed" in dr["assignment"]["mode"] and "filter" in dr["assignment"]["mode"], dr["assignment"]["mode"]
    assert all(x.startswith("exact (built") if tied else x.startswith("filter") for x in modes), modes


def test_float32_storage_of_the_filter_and_the_float32_passes_agree_with_the_float64_ones(g):
    """pm_chi2_filter4_f32 == float32(pm_chi2_filter4) entry by entry, and the float32 forms of the solver's three streaming passes
    (row_select, col_min, certificate) answer a float32 matrix as the float64 forms answer the same values converted."""
    n, m = 1500, 2100
    mv, fx, _ = synth_pair(max(n, m), 71)
    be = g.P.GpuBackend()
    sc_m, sc_f, _ = g.P.build_descriptors(be, be.cloud(np.ascontiguousarray(mv[:, :n])), be.cloud(np.ascontiguousarray(fx[:, :m])))
    F64 = g.K.chi2_filter4(sc_m[0], sc_f[0])
    F32 = g.K.chi2_filter4(sc_m[0], sc_f[0], dtype=g.t.float32)
    assert F32.dtype == g.t.float32 and g.t.equal(F32, F64.to(g.t.float32))
    for t in range(4):
        assert g.t.equal(g.K.chi2_filter_pair(sc_m[0], sc_f[0], t, dtype=g.t.float32), F32[t])
        assert g.t.equal(g.K.chi2_filter_pair(sc_m[0], sc_f[0], t), F64[t])
    A, B = g.L.DeviceMatrix(F32[1]), g.L.DeviceMatrix(F32[1].to(g.t.float64))
    assert A.f32 and not B.f32
    v = B.col_min()
    assert np.array_equal(A.col_min(), v)
    ca, xa, fa = A.row_select(v, 16)
    cb, xb, fb = B.row_select(v, 16)
    assert np.array_equal(ca, cb) and np.array_equal(xa, xb) and fa == fb == 0
    u = xb[:, 0] - v[cb[:, 0]]
    c4r = np.arange(n, dtype=np.int32)
    ra, rb = A.certificate(u, v, c4r, 1.0, 1e-3, 64 * m), B.certificate(u, v, c4r, 1.0, 1e-3, 64 * m)
    assert ra[0] == rb[0] and ra[1] == rb[1]
    ka = np.lexsort((ra[2][:, 1], ra[2][:, 0]))
    kb = np.lexsort((rb[2][:, 1], rb[2][:, 0]))
    assert np.array_equal(ra[2][ka], rb[2][kb]) and np.array_equal(ra[3][ka], rb[3][kb])


def test_filter_mode_leases_what_it_writes_and_stays_resident_where_only_that_fits(g, monkeypatch):
    """the filter route needs four float32 matrices + (at the worst) one pairing's two exact ones — half the
    exact mode's eight float64 matrices — and must lease exactly what it writes.  With the device's free memory made to look like
    the window where that half fits and the eight matrices do not (62k..87k nuclei on a 288 GB device; here 9 000 nuclei against a
    pretended 3.5 GB), the registration stays resident (no streaming, no out-of-memory), keeps a 16 N M-byte buffer and returns
    the exact mode's registration."""
    n = 9000
    mv, fx, _ = synth_pair(n, 21)
    monkeypatch.setattr(g.P, "COST_CACHE_MIN_BYTES", 1 << 20)
    g.P.release_cost_buffers()
    kw = dict(ransac_trials=300, icp_iterations=5, seed=3)
    de, dr = {}, {}
    a = g.P.estimate_transform(mv, fx, details=de, cost_mode='exact', options={"keep_cost_buffer": False}, **kw)
    pretended = 3.5e9
    assert 32.0 * n * n <= 0.85 * pretended < 64.0 * n * n
    monkeypatch.setattr(g.P.GpuBackend, "free_bytes", lambda self: pretended)
    b = g.P.estimate_transform(mv, fx, details=dr, **kw)
    assert dr["assignment"]["cost_mode"] == "filter" and "streamed" not in str(dr["assignment"].get("mode", ""))
    assert g.P.kept_cost_bytes(g.t.device("cuda", g.t.cuda.current_device())) == 16 * n * n
    for h in range(8):
        assert np.array_equal(de["lsa"][h][1], dr["lsa"][h][1]), h
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # and below the window even the filter streams, pairing by pairing, with the same answer
    monkeypatch.setattr(g.P.GpuBackend, "free_bytes", lambda self: 2.0e9)
    g.P.release_cost_buffers()
    ds = {}
    c = g.P.estimate_transform(mv, fx, details=ds, **kw)
    assert "streamed" in str(ds["assignment"].get("mode", "")), ds["assignment"].get("mode")
    assert np.array_equal(a[2], c[2]) and np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1])
    g.P.release_cost_buffers()


def test_banded_tile_launches_write_the_single_launch_matrices(g, monkeypatch):
    """A launch holds fewer than 2^32 work-items: from ~130 000 x 130 000 nuclei the tile launchers (16 x 64 entries per workgroup
    of 256) cut the tile rows into bands, one launch per band on offset pointers (csrc/pm_chi2.hip: band_tile_rows; found at
    140 000 nuclei, where one launch left most of the filter matrix unwritten).  PM_CHI2_MAX_BLOCKS forces the banding at a size
    the suite can afford — ragged last band, ragged last tile row — and every build must write the bits of its one-launch self:
    the exact eight (table kernel and generic four-frame kernel), one exact pairing, the relaxed build, the filter in both
    storage types, four pairings at once and one alone."""
    n, m = 1111, 777                                    # 70 tile rows (the last one 7 rows) x 13 tile columns
    mv, fx, _ = synth_pair(max(n, m), 5)
    be = g.P.GpuBackend()
    sc_m, sc_f, _ = g.P.build_descriptors(be, be.cloud(np.ascontiguousarray(mv[:, :n])), be.cloud(np.ascontiguousarray(fx[:, :m])))

    def builds():
        out = {"exact8": g.K.chi2_cost8(sc_m, sc_f), "general8": g.K.chi2_cost8(sc_m, sc_f, path="general"),
               "computed8": g.K.chi2_cost8(sc_m, sc_f, path="symmetric-computed"),
               "relaxed": g.K.chi2_cost8_relaxed(sc_m[0], sc_f[0]), "filter4": g.K.chi2_filter4(sc_m[0], sc_f[0]),
               "filter4_f32": g.K.chi2_filter4(sc_m[0], sc_f[0], dtype=g.t.float32), "one": g.K.chi2_cost(sc_m[1], sc_f[2])}
        for t in range(4):
            out["pair%d" % t] = g.K.chi2_cost_pair(sc_m, sc_f, t, True)
            out["filter_pair%d" % t] = g.K.chi2_filter_pair(sc_m[0], sc_f[0], t, dtype=g.t.float32)
        g.t.cuda.synchronize()
        return out

    monkeypatch.delenv("PM_CHI2_MAX_BLOCKS", raising=False)
    whole = builds()
    for cap in (13, 13 * 9 + 5, 13 * 69):               # one tile row per band (70 launches); 9 per band, ragged; 69 + 1
        monkeypatch.setenv("PM_CHI2_MAX_BLOCKS", str(cap))
        banded = builds()
        for k in whole:
            a, b = whole[k], banded[k]
            if isinstance(a, (tuple, list)):
                assert all(g.t.equal(x, y) for x, y in zip(a, b)), (cap, k)
            else:
                assert g.t.equal(a, b), (cap, k)
    monkeypatch.setenv("PM_CHI2_MAX_BLOCKS", "12")      # narrower than one tile row: refused, not truncated
    with pytest.raises(Exception):
        g.K.chi2_filter_pair(sc_m[0], sc_f[0], 0, dtype=g.t.float32)
