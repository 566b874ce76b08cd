"""BASELINE config 5 at its stated sizes on the HIP path: a batch of independent specimen pairs of mixed sizes (2k-6k nuclei
here; tools/batch_throughput.py runs the 64 pairs of 2k-20k), each a complete unsupervised registration following
_dock_widget.py:526-718 — batch == stand-alone calls bit for bit, and the smallest pair checked stage by stage against
the CPU oracle (Hungarian indices, RANSAC inlier counts, every ICP iteration's nearest neighbours, final 4x4)."""
import numpy as np
import pytest

from conftest import synth_pair

pytestmark = pytest.mark.gpu

SIZES = [(2000, 2000), (6000, 5800), (3500, 3500), (2500, 2700), (4800, 4800), (3000, 2900), (5500, 5500), (2200, 2200)]
KW = dict(ransac_trials=1500, ransac_error=16, icp_iterations=25)


@pytest.fixture(scope="module")
def pairs():
    import torch
    from platymatch_amd import _native as nat
    from platymatch_amd.build import build_native
    from platymatch_amd.estimate_transform import perform_icp as pi
    build_native()
    nat.load()
    assert torch.cuda.is_available()
    pi.VERBOSE = False
    out = []
    for k, (n, m) in enumerate(SIZES):
        mv, fx, A = synth_pair(max(n, m), 500 + k)
        out.append((np.ascontiguousarray(mv[:, :n]), np.ascontiguousarray(fx[:, :m]), A))
    return out


def test_batch_of_eight_pairs_2k_to_6k_equals_sequential_and_recovers_the_transform(pairs):
    from platymatch_amd import pipeline as P
    seeds = [11 * k + 3 for k in range(len(pairs))]
    seq = [P.estimate_transform(a, b, seed=s, **KW) for (a, b, _), s in zip(pairs, seeds)]
    timings = {}
    par = P.estimate_transform_batch([(a, b) for a, b, _ in pairs], workers=4, seeds=seeds, timings=timings, **KW)
    assert sorted(timings) == list(range(len(pairs))) and all("host_assignment" in t for t in timings.values())
    for k, ((s_sc, s_icp, s_inl), (p_sc, p_icp, p_inl)) in enumerate(zip(seq, par)):
        assert np.array_equal(s_inl, p_inl) and np.array_equal(s_sc, p_sc) and np.array_equal(s_icp, p_icp), k
        A = pairs[k][2]
        # sigma = 1 jitter on the fixed cloud: the ground-truth transform is recovered to the noise level
        assert np.linalg.norm(p_icp @ p_sc - A) / np.linalg.norm(A) < 2e-3, k
        assert p_inl.max() >= 0.9 * min(SIZES[k])


def test_smallest_pair_of_the_batch_matches_the_oracle_stage_by_stage(pairs, oracle):
    from platymatch_amd import pipeline as P
    k = int(np.argmin([n * m for n, m in SIZES]))
    mv, fx, _ = pairs[k]
    kw = dict(KW, ransac_trials=600, icp_iterations=6)
    det_g, det_o = {}, {}
    A_sc, A_icp, inl = P.estimate_transform(mv, fx, seed=5, details=det_g, **kw)
    o_sc, o_icp, o_inl = oracle.estimate_transform(mv, fx, seed=5, details=det_o, **kw)
    for h in range(8):
        assert np.array_equal(det_g["lsa"][h][0], det_o["lsa"][h][0]) and np.array_equal(det_g["lsa"][h][1], det_o["lsa"][h][1]), h
    assert np.array_equal(inl, o_inl)
    assert np.array_equal(det_g["nn"], det_o["nn"])
    ref = o_icp @ o_sc
    assert np.linalg.norm(A_icp @ A_sc - ref) / np.linalg.norm(ref) < 1e-9


def test_config5_at_its_stated_size_64_pairs_of_2k_to_20k(pairs):
    del pairs                                     # (the module fixture only makes sure the library is built and loaded)
    """BASELINE config 5 as stated: 64 specimen pairs, sizes default_rng(5).integers(2000, 20001) (SURVEY.md §8d), each a complete
    unsupervised registration (_dock_widget.py:526-718: 8 x 8 000 RANSAC trials, 50 ICP iterations), one GPU, eight worker
    threads with a HIP stream each.  Every one of the 512 assignments must come from the device-resident route with a
    certificate (no dense host solve), every pair must recover its ground-truth transform to the noise level, the three
    largest pairs must equal their stand-alone registrations bit for bit, and LP duality is re-checked on the host for the
    winning hypothesis of the largest pair.  Prints registrations/s for the seeded (NumPy-stream draws) and the unseeded
    (device sampler) batch."""
    import time
    import torch
    from platymatch_amd import _kernels as K, lsap as L, pipeline as P
    from platymatch_amd.estimate_transform import perform_icp as pi
    pi.VERBOSE = False
    sizes = [int(x) for x in np.random.default_rng(5).integers(2000, 20001, size=64)]
    assert min(sizes) >= 2000 and max(sizes) <= 20000
    batch, truth = [], []
    for k, n in enumerate(sizes):
        mv, fx, A = synth_pair(n, 100 + k)
        batch.append((mv, fx))
        truth.append(A)
    kw = dict(ransac_trials=8000, ransac_error=16, icp_iterations=50)
    P.estimate_transform(batch[0][0][:, :1500], batch[0][1][:, :1500], ransac_trials=100, icp_iterations=2)      # warm-up
    seeds = list(range(64))
    reports = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = P.estimate_transform_batch(batch, workers=8, seeds=seeds, reports=reports, cost_mode='exact', **kw)
    torch.cuda.synchronize()
    dt_seeded = time.perf_counter() - t0
    assert sorted(reports) == list(range(64))
    routes = [r for k in range(64) for r in reports[k]["routes"]]
    assert len(routes) == 512 and all(r is not None and r.startswith("device") and "near-tie" not in r for r in routes), sorted(set(map(str, routes)))
    worst = 0.0
    for k, ((A_sc, A_icp, inl), A) in enumerate(zip(out, truth)):
        err = np.linalg.norm(A_icp @ A_sc - A) / np.linalg.norm(A)
        worst = max(worst, err)
        assert err < 2e-3, (k, sizes[k], err)                   # sigma = 1 jitter on the fixed cloud
        assert inl.max() >= 0.9 * sizes[k], (k, inl)
    # the three largest pairs on their own: identical bits
    big3 = sorted(range(64), key=lambda k: -sizes[k])[:3]
    det = {}
    for k in big3:
        d = det if k == big3[0] else None
        A_sc, A_icp, inl = P.estimate_transform(batch[k][0], batch[k][1], seed=seeds[k], details=d, **kw)
        assert np.array_equal(inl, out[k][2]) and np.array_equal(A_sc, out[k][0]) and np.array_equal(A_icp, out[k][1]), k
    # LP duality for the winning hypothesis of the largest pair, re-checked on the host in extended precision
    k = big3[0]
    h = int(np.argmax(out[k][2]))
    be = P.GpuBackend()
    mov, fix = be.cloud(batch[k][0]), be.cloud(batch[k][1])
    sc_m, sc_f, _ = P.build_descriptors(be, mov, fix)
    name = P.HYPOTHESES[h]
    U = K.chi2_cost(sc_m[int(name[0]) - 1], sc_f[int(name[1]) - 1])
    M = L.DeviceMatrix(U)
    info = {}
    sol = L.solve_core(M, info)
    assert sol is not None and L.certify(M, *sol, info=info), info
    u, v, c4r = sol
    n = sizes[k]
    assert np.array_equal(c4r.astype(np.int64), det["lsa"][h][1]) and np.array_equal(np.sort(c4r), np.arange(n))
    assigned = U[torch.arange(n, device=U.device), torch.as_tensor(c4r.astype(np.int64), device=U.device)].cpu().numpy()
    primal = np.sum(assigned.astype(np.longdouble))
    dual = np.sum(u.astype(np.longdouble)) + np.sum(v.astype(np.longdouble))
    assert abs(float(primal - dual)) <= 1e-9 * float(primal), (float(primal), float(dual))
    del U, M
    # the default for callers who do not seed: index sets drawn on the device
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out_u = P.estimate_transform_batch(batch, workers=8, **kw)
    torch.cuda.synchronize()
    dt_unseeded = time.perf_counter() - t0
    worst_u = max(np.linalg.norm(o[1] @ o[0] - A) / np.linalg.norm(A) for o, A in zip(out_u, truth))
    assert worst_u < 2e-3
    # THE DEFAULT, cost_mode='auto' (pairs below 8 192 nuclei: solved on relaxed matrices; from 8 192: through the float32 filter;
    # both certified on the exact matrices' listed entries): the same batch, seeded as above — every result identical to the exact
    # mode's, bit for bit
    reports_r = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out_r = P.estimate_transform_batch(batch, workers=8, seeds=seeds, reports=reports_r, **kw)
    torch.cuda.synchronize()
    dt_auto = time.perf_counter() - t0
    for k in range(64):
        assert np.array_equal(out_r[k][2], out[k][2]) and np.array_equal(out_r[k][0], out[k][0]) and np.array_equal(out_r[k][1], out[k][1]), k
    modes = [m for k in range(64) for m in reports_r[k]["cost_modes"]]
    assert len(modes) == 512 and all(m and (m.startswith("relaxed") or m.startswith("filter") or m.startswith("exact (")) for m in modes)
    for k in range(64):
        first = "filter" if sizes[k] >= P.FILTER_MIN_POINTS else "relaxed"
        assert all(m.startswith(first) or m.startswith("exact (") for m in reports_r[k]["cost_modes"]), (k, sizes[k], reports_r[k]["cost_modes"])
    # exact costs for callers who do not seed, for the record of what the default buys (device sampler)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out_x = P.estimate_transform_batch(batch, workers=8, cost_mode='exact', **kw)
    torch.cuda.synchronize()
    dt_exact_unseeded = time.perf_counter() - t0
    assert max(np.linalg.norm(o[1] @ o[0] - A) / np.linalg.norm(A) for o, A in zip(out_x, truth)) < 2e-3
    print("\nconfig 5, default cost_mode='auto' (seeded): %.2f s = %.2f registrations/s; of 512 assignments %d settled on the relaxed build, "
          "%d through the filter, %d after an exact build; all 64 results identical to cost_mode='exact' (%.2f s = %.2f registrations/s)"
          % (dt_auto, 64 / dt_auto, sum(m.startswith("relaxed") for m in modes), sum(m.startswith("filter") for m in modes),
             sum(m.startswith("exact") for m in modes), dt_seeded, 64 / dt_seeded))
    print("\nconfig 5 unseeded: default %.2f s = %.2f registrations/s; cost_mode='exact' %.2f s = %.2f registrations/s"
          % (dt_unseeded, 64 / dt_unseeded, dt_exact_unseeded, 64 / dt_exact_unseeded))
    print("\nconfig 5: 64 pairs of %d..%d nuclei on one GPU: seeded (NumPy-stream draws) %.2f s = %.2f registrations/s; unseeded "
          "(device sampler) %.2f s = %.2f registrations/s; worst rel. error vs ground truth %.1e / %.1e; 512 of 512 assignments "
          "device-certified; primal - dual of the largest pair's winner %.1e"
          % (min(sizes), max(sizes), dt_seeded, 64 / dt_seeded, dt_unseeded, 64 / dt_unseeded, worst, worst_u, float(primal - dual)))


def test_two_threads_on_one_stream_never_share_a_kept_cost_buffer(monkeypatch):
    """the kept cost buffer of a (device, stream) is LEASED — a second registration arriving on the same
    stream while the first still reads its matrices (host-driven assignment passes) gets a fresh allocation, never a view of the
    same storage.  With the threshold lowered so that 1 500-point pairs take the kept-buffer path, three threads registering
    different pairs on the default stream at once return exactly what the same calls return one after the other."""
    import threading
    import torch
    from platymatch_amd import pipeline as P
    from platymatch_amd.estimate_transform import perform_icp as pi
    pi.VERBOSE = False
    monkeypatch.setattr(P, "COST_CACHE_MIN_BYTES", 1 << 20)
    P.release_cost_buffers()
    pairs = []
    for k, n in enumerate((1500, 1400, 1600)):
        mv, fx, _ = synth_pair(n, 300 + k)
        pairs.append((mv, fx))
    kw = dict(ransac_trials=200, icp_iterations=4)
    want = [P.estimate_transform(a, b, seed=9 + k, **kw) for k, (a, b) in enumerate(pairs)]
    assert P.kept_cost_bytes(torch.device("cuda", torch.cuda.current_device())) > 0          # the path under test is the kept buffer's
    for rep in range(3):
        got, errors = [None] * 3, []

        def run(k):
            try:
                got[k] = P.estimate_transform(pairs[k][0], pairs[k][1], seed=9 + k, options={"private_rng": True}, **kw)
            except BaseException as e:
                errors.append(e)
        th = [threading.Thread(target=run, args=(k,)) for k in range(3)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errors, errors
        for k in range(3):
            assert np.array_equal(got[k][2], want[k][2]) and np.array_equal(got[k][0], want[k][0]) and np.array_equal(got[k][1], want[k][1]), (rep, k)
    P.release_cost_buffers()
