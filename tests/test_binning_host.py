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


# ---- the kernel's fused four-frame step (pm_bin_index4: ring thresholds, one-shot phi classification + exact fallback)
SIGNS = [(1.0, 1.0), (-1.0, -1.0), (1.0, -1.0), (-1.0, 1.0)]     # frames 1..4: (x, y) sign flips (shape_context.py:172-181)


def prod_bin4(lib, nb, md, nframes=4):
    nb = np.ascontiguousarray(nb, dtype=np.float64)
    out = np.empty((len(nb), 4), np.int32)
    lib.pmt_bin_index4(nb.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(len(nb)), ctypes.c_double(md), ctypes.c_int(nframes),
                       out.ctypes.data_as(ctypes.c_void_p))
    return out


def oracle_bin4(oracle, nb, md):
    return np.stack([oracle_bin(oracle, nb * np.array([sx, sy, 1.0]), md) for sx, sy in SIGNS], axis=1)


def test_four_frame_step_random_and_edges(host_lib, oracle, micro):
    rng = np.random.default_rng(2)
    nb = rng.normal(size=(1_000_000, 3)) * rng.uniform(0.1, 100, size=(1_000_000, 1))
    for md in (37.0, 114.13353403330422, 1.0):
        assert np.array_equal(prod_bin4(host_lib, nb, md), oracle_bin4(oracle, nb, md))
    assert np.array_equal(prod_bin4(host_lib, nb[:1000], 37.0, nframes=2)[:, :2], oracle_bin4(oracle, nb[:1000], 37.0)[:, :2])
    assert (prod_bin4(host_lib, nb[:1000], 37.0, nframes=2)[:, 2:] == -1).all()
    for md in (1.0, 3.0, 0.7):
        assert np.array_equal(prod_bin4(host_lib, micro["grid_neighbors"], md), oracle_bin4(oracle, micro["grid_neighbors"], md))
    tests = []
    for m in range(13):
        for k in list(range(-40, 41)) + [-10 ** 3, 10 ** 3, -10 ** 5, 10 ** 5]:      # ulps around each edge, then just outside the margin
            ang = m * np.pi / 6 + (k * 2.0 ** -52 if abs(k) > 40 else 0.0)
            if abs(k) <= 40:
                for _ in range(abs(k)):
                    ang = math.nextafter(ang, math.inf if k > 0 else -math.inf)
            for rad in (1.0, 3.7, 1e-3, 123.456):
                for zz in (0.0, 0.5, -2.0):
                    tests.append((rad * math.cos(ang), rad * math.sin(ang), zz))
                    tests.append((rad * math.sin(ang), zz, rad * math.cos(ang)))
    # points at relative distance ~2^-40 from the 30/60 degree rays and the axes: both sides of the safety margin
    for t in (math.tan(math.pi / 6), math.tan(math.pi / 3)):
        for eps in (0.0, 2.0 ** -41, 2.0 ** -40, 2.0 ** -39, -2.0 ** -41, -2.0 ** -39, 1e-9, -1e-9):
            for sx in (1, -1):
                for sy in (1, -1):
                    tests.append((sx * 10.0, sy * 10.0 * t * (1 + eps), 1.0))
    for eps in (2.0 ** -41, 2.0 ** -39, 1e-300, 1e-20):
        tests += [(10.0, eps * 10, 1.0), (10.0, -eps * 10, 1.0), (-10.0, eps * 10, 1.0), (eps * 10, 10.0, 1.0), (-eps * 10, -10.0, 1.0)]
    tests = np.array(tests)
    assert np.array_equal(prod_bin4(host_lib, tests, 1.0), oracle_bin4(oracle, tests, 1.0))
    z = [0.0, -0.0, 1.0, -1.0, 1e-300, -1e-300, 1e-310, 1e150, -1e153, np.nan]
    sp = np.array([(x, y, w) for x in z for y in z for w in z])
    a, b = prod_bin4(host_lib, sp, 1.0), oracle_bin4(oracle, sp, 1.0)
    ratio_underflow = (np.abs(sp[:, 1]) > 0) & (np.abs(sp[:, 1]) < np.abs(sp[:, 0]) * 1e-300)
    assert np.array_equal(a[~ratio_underflow], b[~ratio_underflow])


def test_ring_thresholds_equal_the_division(host_lib):
    """#{k : r_ >= rho[k]} must be the reference's r_index for every r_ (rho from pm_ring_thresholds)."""
    edges = np.array([0.125, 0.25000000000000006, 0.5000000000000001, 1.0])
    rng = np.random.default_rng(3)
    for md in [1.0, 3.0, 0.7, 37.0, 114.13353403330422, 1e-3, 12345.678, 1 / 3, float(rng.uniform(1, 200)), 2.0 ** -30, 1e300, 1e-300]:
        rho = np.zeros(4)
        host_lib.pmt_ring_thresholds(ctypes.c_double(md), rho.ctypes.data_as(ctypes.c_void_p))
        cand = [rng.uniform(0, 3 * md, size=20000)]
        for k in range(4):                       # and a dense neighbourhood of every edge
            c = edges[k] * md
            pts = [c]
            for _ in range(50):
                pts.append(math.nextafter(pts[-1], math.inf))
            lo = c
            for _ in range(50):
                lo = math.nextafter(lo, -math.inf)
                pts.append(lo)
            cand.append(np.array(pts))
        r_ = np.concatenate(cand)
        with np.errstate(over="ignore"):
            want = np.where((r_ / md)[:, None] < edges[None, :], 1, 0)
        want_idx = np.where(want.any(1), want.argmax(1), 4)
        got_idx = (r_[:, None] >= rho[None, :]).sum(1)
        assert np.array_equal(got_idx, want_idx), md
    for md in (0.0, float("nan")):               # degenerate mean distance: `r < edge` never holds -> ring 4
        rho = np.zeros(4)
        host_lib.pmt_ring_thresholds(ctypes.c_double(md), rho.ctypes.data_as(ctypes.c_void_p))
        assert (np.array([0.0, 1.0, 1e300])[:, None] >= rho[None, :]).all()
    rho = np.zeros(4)
    host_lib.pmt_ring_thresholds(ctypes.c_double(float("inf")), rho.ctypes.data_as(ctypes.c_void_p))
    assert not (np.array([0.0, 1.0, 1e300])[:, None] >= rho[None, :]).any()      # r = 0 for every finite r_: ring 0


def test_four_frame_step_ring_and_theta_decided_on_squares(host_lib, oracle):
    """pm_bin_index4 decides the ring and the theta step on s = |v|^2 when s is clear of every step by 2^-40 and by the
    reference's sqrt / division otherwise: neighbours at 0, a few ulps, 2^-41, 2^-40, 2^-39 and 1e-9 (relative) on both
    sides of every ring edge and every theta step, for several mean distances, directions and magnitudes."""
    rels = [0.0, 2.0 ** -52, -2.0 ** -52, 3 * 2.0 ** -52, -3 * 2.0 ** -52, 2.0 ** -45, -2.0 ** -45, 2.0 ** -41, -2.0 ** -41,
            2.0 ** -40, -2.0 ** -40, 1.5 * 2.0 ** -40, -1.5 * 2.0 ** -40, 2.0 ** -39, -2.0 ** -39, 1e-9, -1e-9]
    rng = np.random.default_rng(8)
    for md in (1.0, 37.0, 114.13353403330422, 0.7, 1e-3, 12345.678):
        tests = []
        for edge in (0.125, 0.25000000000000006, 0.5000000000000001, 1.0, 2.0):
            for rel in rels:
                rad = edge * md * (1.0 + rel)
                for _ in range(6):
                    u = rng.normal(size=3)
                    u /= np.linalg.norm(u)
                    tests.append(tuple(rad * u))
                tests += [(rad, 0.0, 0.0), (0.0, -rad, 0.0), (0.0, 0.0, rad), (0.0, 0.0, -rad)]
        for k in range(0, 7):                         # theta steps at k * pi / 6
            for rel in rels:
                th = k * np.pi / 6 + rel * (1.0 if k else 0.0) + (abs(rel) if k == 0 else 0.0)
                for rad in (0.3 * md, 1.7 * md, 1e-4 * md, 4e3 * md):
                    for ph in (0.1, 2.0, 4.4, 6.0):
                        tests.append((rad * np.sin(th) * np.cos(ph), rad * np.sin(th) * np.sin(ph), rad * np.cos(th)))
        tests = np.array(tests)
        assert np.array_equal(prod_bin4(host_lib, tests, md), oracle_bin4(oracle, tests, md)), md
