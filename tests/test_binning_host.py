"""The product's binning header (platymatch_amd/csrc/pm_binning.h — thresholds, no acos/atan2) compiled
for the host and compared with the oracle's libm-based bin index.  CPU only."""
import ctypes
import math
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def host_lib():
    out_dir = os.path.join(ROOT, "tests", "csrc", "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libbinning_host.so")
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-ffp-contract=off", "-std=gnu11", "-Wno-unknown-pragmas",
                           "-I" + os.path.join(ROOT, "platymatch_amd", "csrc"), "-shared", "-o", so,
                           os.path.join(ROOT, "tests", "csrc", "binning_host.c"), "-lm"])
    return ctypes.CDLL(so)


def prod_bin(lib, nb, md):
    nb = np.ascontiguousarray(nb, dtype=np.float64)
    out = np.empty(len(nb), np.int32)
    lib.pmt_bin_index(nb.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(len(nb)), ctypes.c_double(md), out.ctypes.data_as(ctypes.c_void_p))
    return out


def oracle_bin(oracle, nb, md):
    with np.errstate(all="ignore"):
        v = oracle.get_bin_index_direct(nb, md)
        ok = (v >= 0) & (v < 360) & (v == np.floor(v))
        return np.where(ok, np.nan_to_num(v, nan=-1), -1).astype(np.int32)


def test_lattice_edges(host_lib, oracle, micro):
    for md in (1.0, 3.0, 0.7, 2.0):
        assert np.array_equal(prod_bin(host_lib, micro["grid_neighbors"], md), oracle_bin(oracle, micro["grid_neighbors"], md))


def test_random_vectors(host_lib, oracle):
    rng = np.random.default_rng(1)
    nb = rng.normal(size=(2_000_000, 3)) * rng.uniform(0.1, 100, size=(2_000_000, 1))
    assert np.array_equal(prod_bin(host_lib, nb, 37.0), oracle_bin(oracle, nb, 37.0))


def test_within_ulps_of_every_edge(host_lib, oracle):
    tests = []
    for m in range(13):
        for k in range(-40, 41):
            ang = m * np.pi / 6
            for _ in range(abs(k)):
                ang = math.nextafter(ang, math.inf if k > 0 else -math.inf)
            for rad in (1.0, 3.7, 1e-3, 123.456, 0.3333333):
                for zz in (0.0, 0.5, -2.0):
                    tests.append((rad * math.cos(ang), rad * math.sin(ang), zz))      # phi edges
                    tests.append((rad * math.sin(ang), zz, rad * math.cos(ang)))      # theta edges
    tests = np.array(tests)
    assert np.array_equal(prod_bin(host_lib, tests, 1.0), oracle_bin(oracle, tests, 1.0))
    # ring edges: r exactly on, one ulp below and above each edge of np.logspace(log10(1/8), log10(2), 5)
    e = [0.125, 0.25, 0.25000000000000006, 0.5, 0.5000000000000001, 1.0, 2.0]
    rs = []
    for v in e:
        rs += [math.nextafter(v, 0), v, math.nextafter(v, 9)]
    t = np.array([(r * 0.6, r * 0.8, 0.0) for r in rs] + [(0.0, r * 0.6, r * 0.8) for r in rs])
    assert np.array_equal(prod_bin(host_lib, t, 1.0), oracle_bin(oracle, t, 1.0))


def test_zeros_signs_nan_and_extreme_magnitudes(host_lib, oracle):
    z = [0.0, -0.0, 1.0, -1.0, 1e-300, -1e-300, 5e-324, -5e-324, 1e-310, 1e150, -1e153, np.nan]
    tests = np.array([(x, y, w) for x in z for y in z for w in z])
    a, b = prod_bin(host_lib, tests, 1.0), oracle_bin(oracle, tests, 1.0)
    # documented limit: |y/x| below 2^-1074 makes libm's atan2 underflow to -0 (sector 0) where
    # the exact angle is in the last sector; unreachable with finite-precision cloud coordinates
    ratio_underflow = (np.abs(tests[:, 1]) > 0) & (np.abs(tests[:, 1]) < np.abs(tests[:, 0]) * 1e-300)
    assert np.array_equal(a[~ratio_underflow], b[~ratio_underflow])


def test_tables_regenerate(oracle):
    """pm_bin_tables.h is what gen_bin_tables.py prints (mpmath): guards against a stale or hand-edited table."""
    gen = os.path.join(ROOT, "platymatch_amd", "csrc", "gen_bin_tables.py")
    out = subprocess.run(["python", gen], capture_output=True, text=True, timeout=120, check=True).stdout
    with open(os.path.join(ROOT, "platymatch_amd", "csrc", "pm_bin_tables.h")) as fh:
        assert fh.read() == out
