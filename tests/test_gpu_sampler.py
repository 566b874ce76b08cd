"""The device sampler of unseeded RANSAC runs (pm_ransac_draw / pm_ransac_affine_draw).

do_ransac draws a trial's pairs with np.random.choice(n, min_samples, replace=False) from NumPy's global generator
(shape_context.py:122) and the reference never seeds it: the contract is "min_samples distinct pairs, every subset equally
likely".  Checked here: the kernel's output equals a NumPy restatement of its published algorithm (Philox-4x32-10, Lemire's
bounded integers, Floyd's subset algorithm) word for word; no repeats, range, uniform marginals and pair co-occurrence; the
fused draw+fit launch equals draw followed by the seeded-path launch; the distribution of do_ransac's inlier counts over
200 seeds is indistinguishable from the NumPy-stream path's; seeded calls still take the NumPy stream (fixtures untouched)."""
import numpy as np
import pytest

from conftest import load_golden, synth_pair

pytestmark = pytest.mark.gpu

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
U32 = np.uint64(0xFFFFFFFF)


def philox_block(seed, trial, block, run):
    """Philox-4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11), vectorised over
    trials: counter (trial, block, run, 0), key (seed low, seed high) -> four uint32 words per trial."""
    x = [np.asarray(trial, dtype=np.uint64), np.full_like(trial, block, dtype=np.uint64), np.full_like(trial, run, dtype=np.uint64),
         np.zeros_like(trial, dtype=np.uint64)]
    k0, k1 = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = np.uint64(M0) * x[0], np.uint64(M1) * x[2]
        x = [((p1 >> np.uint64(32)) ^ x[1] ^ k0) & U32, p1 & U32, ((p0 >> np.uint64(32)) ^ x[3] ^ k1) & U32, p0 & U32]
        k0, k1 = (k0 + np.uint64(W0)) & U32, (k1 + np.uint64(W1)) & U32
    return np.stack(x, axis=1)


def draw_reference(n, k, trials, seed, run):
    """The kernel's algorithm trial by trial (scalar control flow, words from philox_block)."""
    words = {}

    def word(t, i):                                   # i-th 32-bit word of trial t's stream
        b = i // 4
        if b not in words:
            words[b] = philox_block(seed, np.arange(trials, dtype=np.uint64), b, run)
        return int(words[b][t, i % 4])

    out = np.empty((trials, k), dtype=np.int32)
    for t in range(trials):
        used = 0
        for q in range(k):
            j = n - k + q
            rng = j + 1
            m = word(t, used) * rng
            used += 1
            lo = m & 0xFFFFFFFF
            if lo < rng:
                thresh = ((1 << 32) - rng) % rng
                while lo < thresh:
                    m = word(t, used) * rng
                    used += 1
                    lo = m & 0xFFFFFFFF
            pick = m >> 32
            if pick in out[t, :q]:
                pick = j
            out[t, q] = pick
    return out


@pytest.fixture(scope="module")
def K():
    import torch
    from platymatch_amd import _kernels, _native as nat
    from platymatch_amd.estimate_transform import perform_icp as pi
    nat.load()
    assert torch.cuda.is_available()
    pi.VERBOSE = False
    return _kernels


@pytest.mark.parametrize("n,k,trials,seed,run", [(50, 4, 3000, 0x0123456789ABCDEF, 0), (331, 4, 2000, 7, 5), (5, 5, 64, 99, 1),
                                                 (1000, 8, 1500, 2 ** 64 - 1, 7), (4, 4, 10, 1, 0), (3_000_000_000 // 2, 4, 500, 12345, 2),
                                                 (97, 13, 400, 31337, 3), (6, 1, 100, 5, 0)])
def test_device_draws_equal_the_published_algorithm(K, n, k, trials, seed, run):
    got = K.ransac_draw(n, k, trials, seed, run).cpu().numpy()
    assert np.array_equal(got, draw_reference(n, k, trials, seed, run))
    assert got.min() >= 0 and got.max() < n
    s = np.sort(got, axis=1)
    assert (np.diff(s, axis=1) > 0).all() if k > 1 else True          # no repeats inside a trial


def test_rejection_path_of_the_bounded_integer_is_exercised_and_exact(K):
    """range just above 2^31: Lemire's threshold is ~2^31, so about half of the first words are rejected — the retry loop
    runs in almost every trial and must still match the restatement."""
    n, k, trials = 2 ** 31 - 1, 4, 300
    got = K.ransac_draw(n, k, trials, 42, 0).cpu().numpy()
    assert np.array_equal(got, draw_reference(n, k, trials, 42, 0))


def test_subsets_are_uniform_marginals_and_pair_cooccurrence(K):
    from scipy.stats import chi2 as chi2_dist
    n, k, trials = 50, 4, 400_000
    s = K.ransac_draw(n, k, trials, 0xC0FFEE, 3).cpu().numpy().astype(np.int64)
    # marginals: every index in trials * k / n sets
    cnt = np.bincount(s.ravel(), minlength=n)
    e = trials * k / n
    # counts of a k-subset sampler: covariance T[(p - p2) I + (p2 - p^2) J] with p = k/n, p2 = k(k-1)/(n(n-1)); on the
    # complement of the all-ones vector the variance is T(p - p2) = e (n-k)/(n-1), so this is chi-square with n-1 dof
    x2 = ((cnt - e) ** 2 / e).sum() * (n - 1) / (n - k)
    assert chi2_dist.sf(x2, n - 1) > 1e-4, x2
    # pairs: every unordered pair in trials * C(k,2) / C(n,2) sets
    pc = np.zeros((n, n), dtype=np.int64)
    for a in range(k):
        for b in range(a + 1, k):
            lo, hi = np.minimum(s[:, a], s[:, b]), np.maximum(s[:, a], s[:, b])
            np.add.at(pc, (lo, hi), 1)
    iu = np.triu_indices(n, 1)
    e2 = trials * (k * (k - 1) / 2) / (n * (n - 1) / 2)
    x2p = ((pc[iu] - e2) ** 2 / e2).sum()
    dof = len(iu[0]) - 1
    assert chi2_dist.sf(x2p, dof) > 1e-4 and chi2_dist.cdf(x2p, dof) > 1e-4, (x2p, dof)
    # position inside a set does not matter for a fit, but every position must reach every index
    for q in range(k):
        assert np.bincount(s[:, q], minlength=n).min() > 0 or q < k - 1     # (Floyd: position q spans [0, n-k+q])
    # different runs / seeds of the same trial give different sets
    other = K.ransac_draw(n, k, 1000, 0xC0FFEE, 4).cpu().numpy()
    assert (other != s[:1000]).any(axis=1).mean() > 0.9


def test_fused_draw_and_fit_equals_draw_then_fit(K):
    import torch
    d = load_golden("insitu02_affine")
    mov = torch.as_tensor(d["moving"][:3].copy(), device="cuda")
    fix = torch.as_tensor(d["fixed"][:3].copy(), device="cuda")
    rows = torch.as_tensor(d["lsa_rows"][0].astype(np.int32), device="cuda")
    cols = torch.as_tensor(d["lsa_cols"][0].astype(np.int32), device="cuda")
    for k in (4, 6):
        smp, A, inl, deg = K.ransac_affine_draw(mov, fix, rows, cols, k, 5000, 2024, 1, 5.0)
        want = K.ransac_draw(rows.numel(), k, 5000, 2024, 1)
        assert torch.equal(smp, want)
        A2, inl2, deg2 = K.ransac_affine(mov, fix, rows, cols, want, 5.0)
        assert torch.equal(inl, inl2) and torch.equal(deg, deg2)
        assert torch.equal(A.view(torch.int64), A2.view(torch.int64))


def test_inlier_count_distribution_matches_the_numpy_stream_path(K):
    """do_ransac on the reference's own clouds (insitu02, the fixture's first hypothesis = the right one): best inlier
    count over 200 differently seeded runs of each sampler.  Same distribution (two-sample KS), same mean to 2 %."""
    from scipy.stats import ks_2samp
    from platymatch_amd.estimate_transform import shape_context as sc
    d = load_golden("insitu02_affine")
    mv, fx = d["moving"][:3][:, d["lsa_rows"][0]], d["fixed"][:3][:, d["lsa_cols"][0]]
    kw = dict(min_samples=4, trials=40, error=4.0)           # few trials: the count of the best trial still varies
    a, b = [], []
    for s in range(200):
        np.random.seed(1000 + s)
        a.append(sc.do_ransac(mv, fx, **kw)[1])              # NumPy's stream (the reference's sets)
        b.append(sc.do_ransac(mv, fx, device_seed=0x9E3779B97F4A7C15 * (s + 1), run=s % 8, **kw)[1])
    a, b = np.array(a), np.array(b)
    assert a.std() > 0 and b.std() > 0
    assert ks_2samp(a, b).pvalue > 1e-3, (a.mean(), b.mean())
    assert abs(a.mean() - b.mean()) < 0.05 * a.mean() + 2 * np.sqrt(a.var() / 200 + b.var() / 200)


def test_seeded_calls_keep_the_numpy_stream_and_unseeded_driver_uses_the_device(K):
    from platymatch_amd import pipeline as P
    from platymatch_amd.estimate_transform import shape_context as sc
    d = load_golden("insitu02_affine")
    # seeded: the reference fixture's inlier counts, bit for bit (NumPy stream)
    _, _, inl = P.estimate_transform(d["moving"], d["fixed"], ransac_trials=int(d["ransac_trials"]), ransac_error=float(d["ransac_error"]),
                                     icp_iterations=2, seed=int(d["ransac_seed"]))
    assert np.array_equal(inl, d["ransac_inliers"])
    # unseeded: device sampler; NumPy's global generator only gives up the 64-bit key -> repeatable after np.random.seed
    calls = []
    orig = sc.draw_ransac_samples
    sc.draw_ransac_samples = lambda *a, **k: (calls.append(a), orig(*a, **k))[1]
    try:
        out = []
        for _ in range(2):
            np.random.seed(77)
            out.append(P.estimate_transform(d["moving"], d["fixed"], ransac_trials=2000, ransac_error=float(d["ransac_error"]), icp_iterations=5))
        state_after = np.random.get_state()[2]
    finally:
        sc.draw_ransac_samples = orig
    assert not calls                                          # no host shuffles at all
    assert np.array_equal(out[0][2], out[1][2]) and np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    A = out[0][1] @ out[0][0]
    assert np.linalg.norm(A - d["A_gt"]) / np.linalg.norm(d["A_gt"]) < 1e-6
    np.random.seed(77)
    np.random.randint(0, 2 ** 32, size=2, dtype=np.uint64)
    assert np.random.get_state()[2] == state_after            # exactly one 2-word draw was consumed
    # explicit choices
    r_np = P.estimate_transform(d["moving"], d["fixed"], ransac_trials=int(d["ransac_trials"]), ransac_error=float(d["ransac_error"]),
                                icp_iterations=2, seed=int(d["ransac_seed"]), options={"sampler": "numpy"})
    assert np.array_equal(r_np[2], d["ransac_inliers"])
    r_dev = [P.estimate_transform(d["moving"], d["fixed"], ransac_trials=500, ransac_error=float(d["ransac_error"]), icp_iterations=2,
                                  seed=9, options={"sampler": "device"}) for _ in range(2)]
    assert np.array_equal(r_dev[0][2], r_dev[1][2]) and np.array_equal(r_dev[0][0], r_dev[1][0])


def test_similar_mode_and_small_sample_counts_draw_on_the_device_too(K):
    from platymatch_amd.estimate_transform import shape_context as sc
    mv, fx, A_gt = synth_pair(400, 3, sigma=0.0)
    # (pair lists are arbitrary here: the cloud paired with itself, identity is the exact answer)
    A3, k3 = sc.do_ransac(mv, mv, min_samples=3, trials=50, error=1.0, device_seed=5)           # < 4 samples: host pinv, device draws
    assert k3 >= 3
    As, ks = sc.do_ransac(mv, mv, min_samples=4, trials=50, error=1.0, transform="Similar", device_seed=5, run=2)
    assert ks == mv.shape[1] and np.allclose(As, np.eye(4), atol=1e-8)
    # module-level switch for the unchanged widget
    sc.SAMPLER = "device"
    try:
        np.random.seed(3)
        a = sc.do_ransac(mv, mv, min_samples=4, trials=100, error=1.0)
        np.random.seed(3)
        b = sc.do_ransac(mv, mv, min_samples=4, trials=100, error=1.0)
    finally:
        sc.SAMPLER = "numpy"
    assert a[1] == b[1] == mv.shape[1] and np.array_equal(a[0], b[0])


def test_a_caller_who_seeds_numpy_itself_gets_the_references_sets_with_sampler_numpy(K):
    """the only way to seed the reference is np.random.seed(s) before its eight do_ransac calls (it draws from NumPy's
    global generator, shape_context.py:122).  estimate_transform(seed=None) with the default sampler='auto' draws on the device and
    does NOT follow that stream (documented: README, INTEGRATION); with sampler='numpy' it does — the same index sets, inlier counts
    and A_sc as estimate_transform(seed=s), i.e. the reference's — and leaves the global generator where the reference would."""
    from platymatch_amd import pipeline as P
    from platymatch_amd.estimate_transform import perform_icp as pi
    pi.VERBOSE = False
    mv, fx, _ = synth_pair(700, 31)
    kw = dict(ransac_trials=150, icp_iterations=3)
    want = P.estimate_transform(mv, fx, seed=1234, **kw)
    state_after_seeded = np.random.get_state()[1].copy()
    np.random.seed(1234)
    got = P.estimate_transform(mv, fx, seed=None, options={'sampler': 'numpy'}, **kw)
    assert np.array_equal(got[2], want[2]) and np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    assert np.array_equal(np.random.get_state()[1], state_after_seeded)         # consumed exactly like the seeded call
    np.random.seed(1234)
    auto = P.estimate_transform(mv, fx, seed=None, **kw)                        # 'auto' without a seed: the device sampler
    assert auto[2].sum() > 0 and np.isfinite(np.asarray(auto[1])).all()
