"""platymatch_amd/csrc/pm_pairwise.h (NumPy's summation order, restated for the device kernels of transform='Similar') compiled
for the host and compared with NumPy itself: np.sum of contiguous vectors, np.mean over the rows of a C-ordered 3 x N array
(what get_similar_transform takes of the moving cloud, find_transform.py:28), np.mean of a vector (the ICP residual,
utils.py:77-88) — bit for bit, for lengths around every structural boundary (8, 128, the halving rule, the 8 192 buffer)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    out_dir = os.path.join(ROOT, "tests", "csrc", "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libpairwise_host.so")
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-ffp-contract=off", "-std=gnu11", "-shared", "-o", so,
                           "-I", os.path.join(ROOT, "platymatch_amd", "csrc"), os.path.join(ROOT, "tests", "csrc", "pairwise_host.c")])
    L = ctypes.CDLL(so)
    L.pmt_pw_sum.restype = ctypes.c_double
    L.pmt_pw_sum.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.pmt_pw_leaves.restype = ctypes.c_int
    L.pmt_pw_leaves.argtypes = [ctypes.c_int]
    return L


def mine(lib, a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return lib.pmt_pw_sum(a.ctypes.data_as(ctypes.c_void_p), len(a))


LENGTHS = (list(range(1, 300)) + [511, 512, 513, 1000, 1023, 1024, 1025, 4095, 4096, 4097, 8191, 8192, 8193, 8199, 8200, 8320, 8321,
                                  12288, 16383, 16384, 16385, 20000, 24577, 50000, 50001, 65536, 100003, 200000])


def test_buffer_size_is_the_default():
    assert np.getbufsize() == 8192          # the chunk length NumPy's reductions run in (pm_pairwise.h: PM_PW_CHUNK)


def test_sum_equals_numpy_for_every_structural_case(lib):
    rng = np.random.default_rng(0)
    for n in LENGTHS:
        for scale in (1.0, 1e6):
            a = rng.normal(size=n) * scale + rng.uniform(-1, 1) * scale
            assert mine(lib, a) == float(np.sum(a)), n
        assert lib.pmt_pw_leaves(n) >= 1
    b = rng.uniform(0, 1, size=50000)                      # all positive: no cancellation hides an order difference
    assert mine(lib, b) == float(np.sum(b))


def test_means_as_get_similar_transform_takes_them(lib):
    rng = np.random.default_rng(1)
    for n in (150, 5000, 8192, 8193, 20000, 50000, 50001):
        m = np.ascontiguousarray(rng.normal(size=(3, n)) * 50 + 200)
        want = np.mean(m, 1, keepdims=True).ravel()                       # C-ordered rows: chunked pairwise
        got = np.array([mine(lib, m[r]) / n for r in range(3)])
        assert np.array_equal(got, want), n
        m4 = np.ascontiguousarray(rng.normal(size=(4, n)) * 50 + 200)[:3]  # the first three rows of a 4 x n array: the same
        assert np.array_equal(np.array([mine(lib, m4[r]) / n for r in range(3)]), np.mean(m4, 1, keepdims=True).ravel()), n
        f = np.asfortranarray(m)                                           # Fortran order (fancy-indexed matches): column after column
        acc = np.zeros(3)
        for j in range(n):
            acc = acc + f[:, j]
        assert np.array_equal(acc / n, np.mean(f, 1, keepdims=True).ravel()), n
        x = rng.normal(size=n) * 3
        assert mine(lib, x) / n == float(np.mean(x)), n                    # the residual's mean
        y = f[0, :] * m[1, :]                                              # a product of a strided and a contiguous view: contiguous result
        assert y.flags["C_CONTIGUOUS"] and mine(lib, y) == float(np.sum(y)), n
