"""The ICP loop in one launch (pm_icp_one_launch: iterations 1 .. iters-1 by persistent workgroups) against
the launch-per-iteration loop (pm_icp) — same reduction tree, same search, hence the same bits: 4 x 4, every residual, every
iteration's nearest neighbours, the moved cloud.  The launch-per-iteration loop is itself pinned to the reference fixtures and
the oracle (test_gpu_parity.py::test_icp_matches_reference, test_gpu_fullsize.py)."""
import threading

import numpy as np
import pytest

import bench
from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    import torch
    from platymatch_amd import _kernels as K, _native as nat
    nat.load()
    assert torch.cuda.is_available()

    class G:
        pass
    G.K, G.nat, G.t, G.dev = K, nat, torch, torch.device("cuda:0")
    return G


def both(g, start, fix, iters):
    out = []
    for one in (True, False):
        w = start.clone()
        st = g.t.zeros(1, dtype=g.t.int32, device=g.dev)
        A, res, nn = g.K.icp(w, fix, iters, want_nn=True, status=st, one_launch=one)
        g.t.cuda.synchronize()
        out.append((A.cpu().numpy(), res.cpu().numpy(), nn.cpu().numpy(), w.cpu().numpy(), int(st.item())))
    return out


@pytest.mark.parametrize("n,iters", [(8, 5), (100, 3), (513, 4), (5000, 50), (20000, 12), (32768, 7), (50000, 30), (50001, 9)])
def test_one_launch_equals_launch_per_iteration(g, n, iters):
    mv, fx, start = bench.synth(n, seed=n % 7)
    fix, st = g.nat.to_dev(fx, dev=g.dev), g.nat.to_dev(start, dev=g.dev)
    (A1, r1, nn1, w1, s1), (A2, r2, nn2, w2, s2) = both(g, st, fix, iters)
    assert s1 == 0 and s2 == 0
    assert np.array_equal(nn1, nn2)
    assert np.array_equal(A1.view(np.uint64), A2.view(np.uint64))
    assert np.array_equal(r1.view(np.uint64), r2.view(np.uint64))
    assert np.array_equal(w1.view(np.uint64), w2.view(np.uint64))
    assert np.isfinite(A1).all() and r1[-1] <= r1[0]


def test_more_moving_than_fixed_points_and_the_reverse(g):
    for n, m in ((9000, 6000), (6000, 9000)):
        mv, fx, start = bench.synth(max(n, m), seed=3)
        fix = g.nat.to_dev(np.ascontiguousarray(fx[:, :m]), dev=g.dev)
        st = g.nat.to_dev(np.ascontiguousarray(start[:, :n]), dev=g.dev)
        (A1, r1, nn1, w1, _), (A2, r2, nn2, w2, _) = both(g, st, fix, 10)
        assert np.array_equal(nn1, nn2) and np.array_equal(A1, A2) and np.array_equal(r1, r2) and np.array_equal(w1, w2)


def test_reference_fixture_through_the_one_launch_loop(g):
    d = load_golden("insitu02_affine")
    from platymatch_amd.estimate_transform import perform_icp as pi
    pi.VERBOSE = False
    from platymatch_amd.estimate_transform.apply_transform import apply_affine_transform
    moved = apply_affine_transform(d["moving"][:3], d["A_sc"])
    log = {}
    A = pi.perform_icp(moved, d["fixed"][:3], int(d["icp_iters"]), "Affine", log=log)
    assert np.array_equal(log["nn"], d["icp_nn"])
    assert np.linalg.norm(A - d["A_icp"]) / np.linalg.norm(d["A_icp"]) < 1e-9
    assert np.allclose(log["residuals"], d["icp_residuals"], rtol=1e-9, atol=1e-9)       # (exact alignment: residuals at rounding level after the first)


def test_planar_cloud_is_flagged_by_the_one_launch_loop_too(g):
    rng = np.random.default_rng(0)
    flat = np.vstack([rng.normal(size=(2, 2000)) * 50, np.full((1, 2000), 7.0)])
    fix = g.nat.to_dev(flat + rng.normal(scale=0.01, size=flat.shape) * np.array([[1], [1], [0]]), dev=g.dev)
    st = g.t.zeros(1, dtype=g.t.int32, device=g.dev)
    g.K.icp(g.nat.to_dev(flat, dev=g.dev), fix, 5, status=st, one_launch=True)
    assert int(st.item()) == 1


def test_concurrent_callers_are_serialised_and_ordinary_kernels_overlap(g):
    """Three threads on their own streams: two run the one-launch loop at the same time (the per-device lock makes them take
    turns: two persistent grids must never share the device), the third keeps the GPU busy with cost-matrix launches.  Every
    result equals the quiet run's bits and nobody times out (status 0)."""
    t = g.t
    mv, fx, start = bench.synth(50000, seed=1)
    fix, st = g.nat.to_dev(fx, dev=g.dev), g.nat.to_dev(start, dev=g.dev)
    quiet = both(g, st, fix, 40)[0]
    results, errors = {}, []
    stop = threading.Event()

    def icp_worker(k):
        try:
            with t.cuda.stream(t.cuda.Stream(device=g.dev)):
                for rep in range(3):
                    w = st.clone()
                    s = t.zeros(1, dtype=t.int32, device=g.dev)
                    A, res, nn = g.K.icp(w, fix, 40, want_nn=True, status=s)
                    t.cuda.current_stream().synchronize()
                    results[(k, rep)] = (A.cpu().numpy(), res.cpu().numpy(), nn.cpu().numpy(), int(s.item()))
        except BaseException as e:
            errors.append(e)

    def noise():
        try:
            with t.cuda.stream(t.cuda.Stream(device=g.dev)):
                a = t.rand((4096, 360), dtype=t.float64, device=g.dev) + 0.1
                while not stop.is_set():
                    g.K.chi2_cost(a, a)
                    t.cuda.current_stream().synchronize()
        except BaseException as e:
            errors.append(e)

    th = [threading.Thread(target=icp_worker, args=(k,)) for k in range(2)] + [threading.Thread(target=noise)]
    for x in th:
        x.start()
    for x in th[:2]:
        x.join()
    stop.set()
    th[2].join()
    assert not errors, errors
    assert len(results) == 6
    for key, (A, res, nn, s) in results.items():
        assert s == 0, key
        assert np.array_equal(A, quiet[0]) and np.array_equal(res, quiet[1]) and np.array_equal(nn, quiet[2]), key


def test_the_library_refuses_a_second_persistent_grid_on_the_same_device(g):
    """include/platymatch_hip.h: at most one pm_icp_one_launch in flight per device — enforced by the library itself (round 4),
    not only by the mirror's lock: while one call's work is still running, a second call (another stream, the C ABI directly)
    returns PM_ERR_UNSUPPORTED and enqueues nothing; once the first has drained the same call is accepted and gives pm_icp's bits."""
    t, nat = g.t, g.nat
    lib = nat.load()
    n = 40000
    mv, fx, start = bench.synth(n, seed=2)
    fix = nat.to_dev(fx, dev=g.dev)
    iters = 400

    def args(w, stream):
        ws = nat.workspace(lib.pm_icp_workspace(n, n), g.dev)
        A = t.empty(16, dtype=t.float64, device=g.dev)
        res = t.empty(iters, dtype=t.float64, device=g.dev)
        return (nat.ptr(w), n, nat.ptr(fix), n, iters, nat.ptr(A), nat.ptr(res), None, None, nat.ptr(ws), ws.numel(), stream.cuda_stream), (A, res, ws)

    s1, s2 = t.cuda.Stream(device=g.dev), t.cuda.Stream(device=g.dev)
    t.cuda.synchronize()
    w1, w2 = nat.to_dev(start, dev=g.dev), nat.to_dev(start, dev=g.dev)
    a1, keep1 = args(w1, s1)
    a2, keep2 = args(w2, s2)
    t.cuda.synchronize()
    rc1 = lib.pm_icp_one_launch(*a1)           # ~400 iterations: several milliseconds of device time
    rc2 = lib.pm_icp_one_launch(*a2)           # asked while the first is in flight
    assert rc1 == 0
    assert rc2 == -4, rc2                      # PM_ERR_UNSUPPORTED
    s1.synchronize()
    t.cuda.synchronize()
    rc3 = lib.pm_icp_one_launch(*a2)           # the flag came back with the first call's completion
    assert rc3 == 0
    s2.synchronize()
    assert np.array_equal(keep1[0].cpu().numpy(), keep2[0].cpu().numpy()) and np.array_equal(keep1[1].cpu().numpy(), keep2[1].cpu().numpy())
    w3 = nat.to_dev(start, dev=g.dev)
    A3, res3, _ = g.K.icp(w3, fix, iters)
    t.cuda.synchronize()
    assert np.array_equal(A3.cpu().numpy().reshape(16), keep1[0].cpu().numpy()) and np.array_equal(res3.cpu().numpy(), keep1[1].cpu().numpy())
