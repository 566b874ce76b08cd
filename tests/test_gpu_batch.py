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
