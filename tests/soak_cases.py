"""Random registration cases for the parity soak (tests/probes/soak_parity.py) and its regression tests
(tests/test_gpu_soak.py): sizes N != M, anisotropy, offset, scale, jitter; every fifth cloud on a half-integer lattice, which
puts neighbours exactly on ring and sector edges, makes distances tie and leaves moved points exactly midway between fixed ones."""
import numpy as np


def make_case(seed, max_points=500):
    rng = np.random.default_rng(1000003 * seed + 17)
    n = int(rng.integers(6, max_points + 1))
    m = int(rng.integers(6, max_points + 1)) if rng.random() < 0.7 else n
    big = max(n, m)
    axes = rng.uniform(5.0, 80.0, size=(3, 1)) * rng.choice([1.0, 1.0, 0.2], size=(3, 1))
    base = rng.normal(size=(3, big)) * axes + rng.uniform(-300.0, 300.0, size=(3, 1))
    th = rng.uniform(-0.6, 0.6, size=3)
    Rz = np.array([[np.cos(th[0]), -np.sin(th[0]), 0], [np.sin(th[0]), np.cos(th[0]), 0], [0, 0, 1]])
    Ry = np.array([[np.cos(th[1]), 0, np.sin(th[1])], [0, 1, 0], [-np.sin(th[1]), 0, np.cos(th[1])]])
    A = (Rz @ Ry) * rng.uniform(0.7, 1.4) + rng.normal(scale=0.03, size=(3, 3))
    t = rng.uniform(-50.0, 50.0, size=(3, 1))
    fx = A @ base + t + rng.normal(scale=rng.choice([0.0, 0.3, 1.0]), size=(3, big))
    mv = base.copy()
    lattice = seed % 5 == 4
    if lattice:                                   # half-integer lattice: exact bin-edge hits, tied distances, a few duplicates
        mv, fx = np.round(mv * 0.1) * 5.0, np.round(fx * 0.1) * 5.0
    fx = fx[:, rng.permutation(big)]
    scale = float(rng.choice([1.0, 1.0, 1e-3, 1e3]))
    return (np.ascontiguousarray(mv[:, :n] * scale), np.ascontiguousarray(fx[:, :m] * scale), lattice,
            "Similar" if seed % 3 == 2 else "Affine", int(rng.integers(0, 2 ** 31)))
