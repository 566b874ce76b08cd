"""ctypes binding of libplatymatch_hip.so (include/platymatch_hip.h) and device plumbing.

There is no CPU fallback: if the library cannot be loaded, or no ROCm device is visible, every
product entry point raises.  torch is used for device memory, streams and (elsewhere)
torch.distributed only; every number is produced by the HIP kernels.
"""
import ctypes
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PM_LIB_PATH: another build of the same library (the sanitized host build of platymatch_amd.build.build_sanitized; diagnostic builds)
LIB_PATH = os.environ.get("PM_LIB_PATH") or os.path.join(_HERE, "libplatymatch_hip.so")
NBINS = 360
ABI_VERSION = 2
ICP_NSUMS = 24
_lib = None
_lock = threading.Lock()

_c_void_p, _c_int, _c_size_t, _c_double = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_double

# name -> (restype, argtypes); mirrors include/platymatch_hip.h one to one
SIGNATURES = {
    "pm_version": (_c_int, []),
    "pm_error_string": (ctypes.c_char_p, [_c_int]),
    "pm_last_hip_error": (_c_int, []),
    "pm_device_alloc": (_c_int, [_c_int, _c_size_t, _c_void_p]),
    "pm_device_free": (_c_int, [_c_int, _c_void_p]),
    "pm_device_memory": (_c_int, [_c_int, _c_void_p, _c_void_p]),
    "pm_clock_probe": (_c_int, [_c_void_p, _c_int, ctypes.c_ulonglong, _c_void_p]),
    "pm_centroid_workspace": (_c_size_t, [_c_int]),
    "pm_centroid": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_void_p, _c_size_t, _c_void_p]),
    "pm_centroid_sequential": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_void_p]),
    "pm_mean_distance_workspace": (_c_size_t, [_c_int]),
    "pm_mean_distance": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_void_p, _c_size_t, _c_void_p]),
    "pm_pca_axis_workspace": (_c_size_t, [_c_int]),
    "pm_pca_axis": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_void_p, _c_size_t, _c_void_p]),
    "pm_pca_components": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_void_p]),
    "pm_cdist": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_size_t, _c_void_p]),
    "pm_label_moments": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p]),
    "pm_legacy_choice": (_c_int, [_c_void_p, _c_void_p, ctypes.c_long, _c_int, ctypes.c_long, _c_void_p]),
    "pm_lsap_solve": (_c_int, [_c_void_p, ctypes.c_long, ctypes.c_long, _c_void_p, _c_void_p]),
    "pm_lsap_row_select": (_c_int, [_c_void_p, _c_int, _c_int, _c_size_t, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "pm_lsap_certificate": (_c_int, [_c_void_p, _c_int, _c_int, _c_size_t, _c_void_p, _c_void_p, _c_void_p, _c_double, _c_double,
                                     _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p]),
    "pm_lsap_col_min_workspace": (_c_size_t, [_c_int, _c_int]),
    "pm_lsap_row_select_f32": (_c_int, [_c_void_p, _c_int, _c_int, _c_size_t, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "pm_lsap_certificate_f32": (_c_int, [_c_void_p, _c_int, _c_int, _c_size_t, _c_void_p, _c_void_p, _c_void_p, _c_double, _c_double,
                                         _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p]),
    "pm_lsap_col_min_f32": (_c_int, [_c_void_p, _c_int, _c_int, _c_size_t, _c_void_p, _c_void_p, _c_size_t, _c_void_p]),
    "pm_chi2_filter4_f32": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_size_t, _c_size_t, _c_void_p, _c_size_t, _c_void_p]),
    "pm_chi2_filter_pair_f32": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_int, _c_int, _c_void_p, _c_size_t, _c_void_p, _c_size_t, _c_void_p]),
    "pm_lsap_col_min": (_c_int, [_c_void_p, _c_int, _c_int, _c_size_t, _c_void_p, _c_void_p, _c_size_t, _c_void_p]),
    "pm_lsap_core_init_duals": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "pm_lsap_core_init_state": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "pm_lsap_bid": (_c_int, [_c_void_p, _c_int, _c_int, _c_size_t, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "pm_lsap_core_create": (_c_void_p, [_c_int, _c_int]),
    "pm_lsap_core_destroy": (None, [_c_void_p]),
    "pm_lsap_core_add": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_void_p]),
    "pm_lsap_core_solve": (_c_int, [_c_void_p]),
    "pm_transpose_f64": (_c_int, [_c_void_p, _c_int, _c_int, _c_size_t, _c_void_p, _c_size_t, _c_void_p]),
    "pm_lsap_core_auction": (_c_int, [_c_void_p, _c_double, _c_double, _c_double, ctypes.c_long, _c_void_p]),
    "pm_lsap_core_auction_resume": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_double, _c_double, _c_double, ctypes.c_long, _c_void_p]),
    "pm_lsap_diagonal": (_c_int, [_c_void_p, _c_int, _c_int, _c_size_t, _c_int, _c_void_p, _c_void_p]),
    "pm_lsap_default_options": (None, [_c_void_p]),
    "pm_lsap_resident_workspace": (_c_size_t, [_c_int, _c_int]),
    "pm_lsap_solve_resident": (_c_int, [_c_void_p, _c_int, _c_int, _c_size_t, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                                        _c_void_p, _c_size_t, _c_void_p]),
    "pm_lsap_certify_resident": (_c_int, [_c_void_p, _c_int, _c_int, _c_size_t, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                                          _c_int, _c_void_p, _c_void_p, _c_size_t, _c_void_p]),
    "pm_lsap_core_reprice": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_void_p, _c_double, _c_void_p]),
    "pm_lsap_core_get": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "pm_lsap_core_column_repairs": (ctypes.c_long, [_c_void_p]),
    "pm_lsap_unique": (_c_int, [_c_int, _c_int, _c_void_p, _c_void_p, _c_double, _c_double, _c_void_p, _c_int]),
    "pm_mean_distance_rows": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_size_t, _c_void_p]),
    "pm_mean_distance_finish": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_void_p]),
    "pm_row_argmin": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_size_t, _c_size_t, _c_void_p, _c_void_p, _c_void_p]),
    "pm_shape_context": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_int,
                                  _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "pm_shape_context_workspace": (_c_size_t, [_c_int]),
    "pm_shape_context_tiled": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_int,
                                        _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_size_t, _c_void_p]),
    "pm_shape_context_neighbors": (_c_int, [_c_void_p, _c_int, _c_double, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "pm_shape_context_neighbors_binned": (_c_int, [_c_void_p, _c_int, _c_double, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_int,
                                                   _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "pm_chi2_cost": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_size_t, _c_void_p]),
    "pm_chi2_cost8": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int,
                               _c_void_p, _c_size_t, _c_size_t, _c_void_p]),
    "pm_chi2_symmetry_check": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int,
                                        _c_void_p, _c_void_p]),
    "pm_chi2_cost8_sym": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_size_t, _c_size_t, _c_void_p]),
    "pm_chi2_cost_pair_sym": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_int, _c_int, _c_void_p, _c_size_t, _c_size_t, _c_void_p]),
    "pm_chi2_sym_workspace_bytes": (_c_size_t, [_c_int, _c_int]),
    "pm_chi2_sym_table_info": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "pm_chi2_relaxed_workspace_bytes": (_c_size_t, [_c_int, _c_int]),
    "pm_chi2_relaxed_delta": (_c_double, []),
    "pm_chi2_filter_workspace_bytes": (_c_size_t, [_c_int, _c_int]),
    "pm_chi2_filter_delta": (_c_double, []),
    "pm_chi2_filter4": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_size_t, _c_size_t, _c_void_p, _c_size_t, _c_void_p]),
    "pm_chi2_filter_pair": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_int, _c_int, _c_void_p, _c_size_t, _c_void_p, _c_size_t, _c_void_p]),
    "pm_chi2_entries_sym": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_int, _c_int, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p]),
    "pm_chi2_cost8_relaxed": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_size_t, _c_size_t, _c_void_p, _c_size_t, _c_int, _c_void_p]),
    "pm_chi2_cost8_sym_ws": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_size_t, _c_size_t, _c_void_p, _c_size_t, _c_void_p]),
    "pm_chi2_cost_pair_sym_ws": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_int, _c_int, _c_void_p, _c_size_t, _c_size_t, _c_void_p,
                                          _c_size_t, _c_void_p]),
    "pm_ransac_affine": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_int, _c_int,
                                  _c_double, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "pm_ransac_draw": (_c_int, [_c_int, _c_int, _c_int, ctypes.c_uint64, ctypes.c_uint32, _c_void_p, _c_void_p]),
    "pm_ransac_affine_draw": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_int, _c_int, _c_int,
                                       ctypes.c_uint64, ctypes.c_uint32, _c_double, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "pm_ransac_score": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_int,
                                 _c_double, _c_void_p, _c_void_p]),
    "pm_apply_affine": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p]),
    "pm_fit_affine_workspace": (_c_size_t, [_c_int]),
    "pm_fit_affine": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_size_t, _c_void_p]),
    "pm_icp_nn_workspace": (_c_size_t, [_c_int, _c_int]),
    "pm_icp_nn": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_size_t, _c_void_p]),
    "pm_icp_grid_workspace": (_c_size_t, [_c_int]),
    "pm_icp_grid_build": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_size_t, _c_void_p]),
    "pm_icp_grid_nn": (_c_int, [_c_void_p, _c_int, _c_int, _c_void_p, _c_size_t, _c_void_p, _c_void_p, _c_void_p]),
    "pm_icp_nn_brute_workspace": (_c_size_t, [_c_int, _c_int]),
    "pm_icp_nn_brute": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_size_t, _c_void_p]),
    "pm_icp_accumulate_workspace": (_c_size_t, [_c_int]),
    "pm_icp_accumulate": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                                   _c_size_t, _c_void_p]),
    "pm_icp_update_workspace": (_c_size_t, [_c_int]),
    "pm_icp_update": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p,
                               _c_void_p, _c_void_p, _c_void_p, _c_size_t, _c_void_p]),
    "pm_icp_apply": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                              _c_size_t, _c_void_p]),
    "pm_get_error_workspace": (_c_size_t, [_c_int]),
    "pm_get_error": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_size_t, _c_void_p]),
    "pm_icp_workspace": (_c_size_t, [_c_int, _c_int]),
    "pm_icp": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                        _c_size_t, _c_void_p]),
    "pm_similar_workspace": (_c_size_t, [_c_int]),
    "pm_similar_moments": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_int, _c_int, _c_void_p, _c_void_p, _c_size_t,
                                    _c_void_p]),
    "pm_similar_apply": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_size_t, _c_void_p]),
    "pm_icp_one_launch": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                                   _c_size_t, _c_void_p]),
}


class NativeError(RuntimeError):
    """libplatymatch_hip.so returned a PM_ERR_* code."""


def load():
    """Load the C-ABI library and declare every prototype.  Raises if it is missing:
    the product has no other implementation to fall back to."""
    global _lib
    with _lock:
        if _lib is None:
            # torch bundles its own HIP runtime (SONAME libamdhip64.so.7, the same as /opt/rocm's).  It must be
            # in the process BEFORE this library is opened, so that both resolve to ONE runtime: loaded the
            # other way round the process ends up with two runtimes and torch's streams/pointers mean nothing
            # to ours (hipErrorNoDevice on the first launch).
            import torch  # noqa: F401
            if not os.path.exists(LIB_PATH):
                raise NativeError(
                    "%s not found. Build it with `python -m platymatch_amd.build` (needs hipcc); "
                    "platymatch_amd has no CPU fallback." % LIB_PATH)
            lib = ctypes.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)   # AttributeError if the .so lacks a declared symbol
                fn.restype = res
                fn.argtypes = args
            if lib.pm_version() != ABI_VERSION:
                raise NativeError("libplatymatch_hip.so ABI version %d, expected %d" % (lib.pm_version(), ABI_VERSION))
            _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        lib = load()
        msg = lib.pm_error_string(rc).decode()
        if rc == -3:
            msg += " [hipError %d]" % lib.pm_last_hip_error()
        if rc == -1:
            raise ValueError("platymatch_hip: " + msg)
        raise NativeError("platymatch_hip: " + msg)


# ------------------------------------------------------------------------------------------- device
def torch_mod():
    import torch
    return torch


def device(dev=None):
    torch = torch_mod()
    if dev is not None and torch.device(dev).type != "cuda":
        return torch.device(dev)   # plumbing only (host-side tensors of a caller-supplied backend); kernels insist on .is_cuda
    if not torch.cuda.is_available():
        raise NativeError("no ROCm device visible: platymatch_amd runs on an AMD GPU only (no CPU fallback)")
    if dev is None:
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device(dev)


def stream_ptr(t=None):
    """hipStream_t of torch's current stream — on the device of tensor `t` if given (kernels must be enqueued on a stream
    of the device that owns their operands), else on the current device."""
    cuda = torch_mod().cuda
    if t is not None and t.is_cuda:
        return cuda.current_stream(t.device).cuda_stream
    return cuda.current_stream().cuda_stream


_SIDE_STREAMS = {}
_SIDE_LOCK = None


def side_stream(device, key, priority=0):
    """A side stream of `device` that PERSISTS across calls, one per `key` (priority -1: its kernels are dispatched ahead of
    normal-priority streams' remaining workgroups).  torch's caching allocator keeps freed blocks per
    stream: a stream created afresh for every registration (or every hypothesis) can reuse nothing, every large buffer is a new
    hipMalloc (22 ms per GB on this pool) until memory runs out and the whole cache is flushed — measured as a batch of 64
    registrations paying ~13 s of allocation, and every second 50 000 x 47 000 registration taking 6 s instead of 1.5 s."""
    global _SIDE_LOCK
    import threading
    torch = torch_mod()
    if _SIDE_LOCK is None:
        _SIDE_LOCK = threading.Lock()
    dev = torch.device(device)
    k = (dev.index if dev.index is not None else torch.cuda.current_device(), key, int(priority))
    with _SIDE_LOCK:
        s = _SIDE_STREAMS.get(k)
        if s is None:
            s = _SIDE_STREAMS[k] = torch.cuda.Stream(device=dev, priority=int(priority))
        return s


def is_torch(x):
    return type(x).__module__.startswith("torch")


def to_dev(x, dtype=None, dev=None):
    """numpy / torch (any device) -> contiguous torch tensor on the GPU."""
    torch = torch_mod()
    dtype = dtype or torch.float64
    if is_torch(x):
        t = x.to(device=device(dev) if not x.is_cuda else x.device, dtype=dtype)
    else:
        t = torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).to(device(dev))
    return t.contiguous()


def like_input(t, ref):
    """Return `t` (a GPU tensor) as the kind of array `ref` was: numpy in -> numpy out."""
    if is_torch(ref):
        return t
    return t.cpu().numpy()


def ptr(t):
    return 0 if t is None else t.data_ptr()


def workspace(nbytes, dev):
    torch = torch_mod()
    return torch.empty(max(int(nbytes), 8), dtype=torch.uint8, device=dev)
