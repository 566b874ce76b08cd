"""The C-ABI library builds for gfx950, loads without a GPU and exports every symbol include/*.h declares.
No compute is called here (there is no GPU in the CPU suite)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib_path():
    from platymatch_amd.build import build_native
    return build_native()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "platymatch_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pm_[a-z0-9_]+)\s*\(", text)))


def test_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), n


def test_python_binding_covers_the_header(lib_path):
    from platymatch_amd import _native
    assert sorted(_native.SIGNATURES) == declared_symbols()
    lib = _native.load()
    assert lib.pm_version() == 2
    assert b"workspace" in lib.pm_error_string(-2)


def header_prototypes():
    """name -> (return type, [parameter types]) from the header text (comments stripped, one declaration per ';')."""
    text = open(os.path.join(ROOT, "include", "platymatch_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(pm_[a-z0-9_]+)\s*\(([^;{}]*)\)\s*;", text):
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3).strip()
        plist = [] if params in ("", "void") else [re.sub(r"\s+", " ", q.strip()) for q in params.split(",")]
        protos[name] = (ret, plist)
    return protos


def c_kind(decl):
    """A C parameter / return declaration -> the ctypes kind the binding must use."""
    d = decl.replace("const ", "").strip()
    if "*" in d:
        return "char_p" if d.startswith("char") else "pointer"
    base = d.split(" ")[0] if " " in d and not d.startswith(("unsigned", "long")) else d
    base = re.sub(r"\b[A-Za-z_][A-Za-z0-9_]*$", "", d).strip() or d          # drop the parameter name
    table = {"int": "i32", "int32_t": "i32", "uint32_t": "u32", "unsigned int": "u32", "size_t": "size", "double": "f64",
             "long": "i64", "long long": "i64", "int64_t": "i64", "uint64_t": "u64", "unsigned long long": "u64", "void": "void"}
    assert base in table, decl
    return table[base]


def ctypes_kind(t):
    if t is None:
        return "void"
    return {ctypes.c_void_p: "pointer", ctypes.c_char_p: "char_p", ctypes.c_int: "i32", ctypes.c_int32: "i32", ctypes.c_uint32: "u32",
            ctypes.c_uint: "u32", ctypes.c_size_t: "size", ctypes.c_double: "f64", ctypes.c_long: "i64", ctypes.c_longlong: "i64",
            ctypes.c_int64: "i64", ctypes.c_uint64: "u64", ctypes.c_ulonglong: "u64", ctypes.c_ulong: "u64"}[t]


def test_python_signatures_match_the_header_prototypes():
    """Every entry point's ctypes signature (platymatch_amd/_native.py) against its prototype in include/platymatch_hip.h:
    same number of parameters, each of the same kind and width (pointer, int32, uint32, int64, uint64, size_t, double) — an
    argument added to one side only would otherwise shift every later argument silently."""
    from platymatch_amd import _native
    protos = header_prototypes()
    assert sorted(protos) == sorted(_native.SIGNATURES)
    same = {"size": "u64", "u64": "u64", "i64": "i64"}            # (size_t and uint64_t travel alike on this ABI)
    for name, (ret, params) in protos.items():
        res, args = _native.SIGNATURES[name]
        assert len(args) == len(params), (name, len(args), params)
        want = [c_kind(q) for q in params]
        have = [ctypes_kind(a) for a in args]
        for k, (w, h) in enumerate(zip(want, have)):
            assert same.get(w, w) == same.get(h, h), (name, k, params[k], h)
        wr, hr = c_kind(ret + " x") if "*" not in ret else c_kind(ret), ctypes_kind(res)
        assert same.get(wr, wr) == same.get(hr, hr) or (wr == "pointer" and hr == "pointer"), (name, ret, hr)


def test_workspace_queries_and_argument_errors(lib_path):
    """Host-side behaviour that needs no device: sizes, and rejection before anything is enqueued."""
    from platymatch_amd import _native
    lib = _native.load()
    assert lib.pm_mean_distance_workspace(1000) == 512           # one float64 per 8 192-element piece of the 499 500 pairs (61), padded to 256 B
    assert lib.pm_icp_workspace(50000, 50000) > 50000 * 4
    assert lib.pm_icp_workspace(0, 10) == 0
    assert lib.pm_centroid(None, 10, None, None, 0, None) == -1
    assert lib.pm_chi2_cost(None, 1, None, 1, None, 1, None) == -1
    assert lib.pm_shape_context(None, 0, 0, 0, None, None, None, 3, None, None, None, None) == -1
    assert lib.pm_icp(None, 5, None, 5, 1, None, None, None, None, None, 0, None) == -1


def test_every_entry_point_rejects_bad_arguments_before_touching_the_device(lib_path):
    """All pointers NULL and all sizes zero -> PM_ERR_INVALID_ARG from every int-returning entry point (nothing is enqueued,
    so this runs without a GPU); a non-NULL call with a missing workspace -> PM_ERR_WORKSPACE; size queries answer 0."""
    from platymatch_amd import _native
    lib = _native.load()
    skipped = {"pm_version", "pm_last_hip_error", "pm_error_string", "pm_lsap_core_create", "pm_lsap_core_destroy",
               "pm_chi2_relaxed_delta", "pm_chi2_filter_delta", "pm_lsap_default_options"}      # (a constant; a void function: NULL is ignored)
    for name, (restype, argtypes) in _native.SIGNATURES.items():
        if name in skipped:
            continue
        args = [None if a is ctypes.c_void_p else (0.0 if a is ctypes.c_double else 0) for a in argtypes]
        got = getattr(lib, name)(*args)
        if restype is ctypes.c_size_t or name.endswith("_workspace"):
            assert got == 0, name
        else:
            assert got == -1, (name, got)
    fake = ctypes.c_void_p(0x1000)               # never dereferenced: the workspace check comes first
    assert lib.pm_mean_distance(fake, 100, fake, None, 0, None) == -2
    assert lib.pm_mean_distance_rows(fake, 100, 0, 1, fake, 0, None) == -2
    assert lib.pm_icp(fake, 10, fake, 10, 3, fake, None, None, None, None, 0, None) == -2
    assert lib.pm_icp_nn(fake, 10, fake, 10, fake, None, None, 0, None) == -2
    assert lib.pm_icp_accumulate(fake, 10, fake, 10, None, fake, fake, None, 0, None) == -2
    assert lib.pm_fit_affine(fake, 10, fake, 10, None, fake, None, None, 0, None) == -2
    assert lib.pm_ransac_affine(fake, 10, fake, 10, None, None, 10, fake, 3, 5, 1.0, fake, fake, None, None) == -4      # < 4 pairs: host pinv
    assert lib.pm_get_error(fake, fake, 10, fake, None, 0, None) == -2
    assert lib.pm_chi2_cost8_sym_ws(fake, 10, fake, 10, fake, 10, 100, None, 0, None) == -2
    assert lib.pm_chi2_cost_pair_sym_ws(fake, 10, fake, 10, 1, fake, 10, 100, fake, 64, None) == -2     # workspace too small
    assert lib.pm_chi2_sym_workspace_bytes(10, 20) == 512 + 3600 + 7200
    assert lib.pm_chi2_cost8_relaxed(fake, 10, fake, 10, fake, 10, 100, None, 0, 1, None) == -2       # workspace missing
    assert 0.0 < lib.pm_chi2_relaxed_delta() < 1e-11
    assert lib.pm_shape_context(fake, 10, 8, 5, fake, fake, fake, 4, fake, None, None, None) == -1     # row block outside the cloud
    assert lib.pm_shape_context(fake, 10, 0, 5, fake, fake, fake, 3, fake, None, None, None) == -1     # 3 frames do not exist
    assert lib.pm_shape_context_tiled(fake, 10, 0, 5, fake, fake, fake, 4, fake, None, None, None, None, 0, None) == -2   # workspace missing
    assert lib.pm_shape_context_tiled(fake, 10, 0, 5, fake, fake, fake, 4, fake, None, None, None, fake, 16, None) == -2   # ... or too small
    assert lib.pm_shape_context_tiled(fake, 10, 8, 5, fake, fake, fake, 4, fake, None, None, None, fake, 1 << 20, None) == -1
    assert lib.pm_ransac_draw(10, 11, 5, 1, 0, fake, None) == -1                                      # more samples than pairs
    assert lib.pm_ransac_affine_draw(fake, 10, fake, 10, None, None, 10, 3, 5, 1, 0, 1.0, fake, fake, fake, None, None) == -4
    assert lib.pm_chi2_cost(fake, 4, fake, 4, fake, 3, None) == -1                                     # ld < columns
    assert lib.pm_row_argmin(fake, 2, 4, 8, 8, 16, fake, None, None) == -1                             # overlapping matrices
    assert lib.pm_mean_distance_rows(fake, 100, 2, 2, fake, 1 << 20, None) == -1                       # offset >= stride


def test_product_refuses_to_run_without_a_gpu(lib_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from platymatch_amd import _native
    from platymatch_amd.utils import utils
    with pytest.raises(_native.NativeError):
        utils.get_centroid(np.zeros((3, 5)), transposed=False)
    import platymatch_amd
    with pytest.raises(_native.NativeError):
        platymatch_amd.register(np.zeros((3, 8)), np.zeros((3, 8)))


def test_no_oracle_import_in_product():
    """The product never imports, links or executes anything under oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "platymatch_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "pm_oracle" not in src and "libpm_oracle" not in src, f
