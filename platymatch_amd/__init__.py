"""platymatch_amd — MI355X-native estimate_transform hot path of PlatyMatch.

Drop-in for the reference's platymatch.estimate_transform.* and platymatch.utils.utils
functions (same names, arguments and array conventions), computed by hand-written HIP kernels
for gfx950 behind a C ABI (include/platymatch_hip.h).  There is no CPU fallback: without the
built library and an AMD GPU every entry point raises.
"""

__version__ = "0.1.0"

# The headless driver is platymatch_amd.estimate_transform.estimate_transform (the sub-package keeps the
# reference's name, so the function lives inside it); `register` is a top-level alias.


def register(moving, fixed, **kwargs):
    """Alias of platymatch_amd.estimate_transform.estimate_transform."""
    from .pipeline import estimate_transform as run
    return run(moving, fixed, **kwargs)


def reserve(n, m=None, cost_mode='auto', device=None):
    """Allocate the cost buffer for registrations of n x m nuclei ahead of the first call (pipeline.reserve): a fresh process
    otherwise pays that allocation inside its first estimate_transform (README: cold and warm numbers)."""
    from .pipeline import reserve as run
    return run(n, m, cost_mode=cost_mode, device=device)


def warm_up(device=None):
    """Pay the once-per-process first-use costs of the assignment stage's selection code (torch loads the code objects of its sorting
    / gathering / indexing kernels at their first launch: 0.3-0.7 s) NOW — e.g. when a plug-in is loaded or detections have been
    read, ahead of the first "Run" (lsap.warm_up; reserve() includes it).  Needs a GPU; a no-op ever after."""
    from . import _native as nat, lsap
    nat.load()
    lsap.warm_up(nat.device(device))


def release_memory(device=None):
    """Give the device memory this package holds without using back to the driver: the kept cost buffers no registration holds,
    the idle matrix blocks (device_memory.py) and torch's idle cache.  The next large registration pays its allocation again."""
    from .device_memory import release_everything_idle
    release_everything_idle(device)


def install_as_platymatch():
    """Register this package's modules under the reference's import paths
    (platymatch.estimate_transform.{shape_context,find_transform,apply_transform,perform_icp},
    platymatch.utils.utils) so code written against the reference — e.g. its napari widget,
    which binds these functions by name at _dock_widget.py:15-21 — picks them up unchanged.
    A no-op for modules the real `platymatch` package has already imported."""
    import importlib
    import sys
    import types
    names = ["estimate_transform", "estimate_transform.shape_context", "estimate_transform.find_transform",
             "estimate_transform.apply_transform", "estimate_transform.perform_icp", "utils", "utils.utils"]
    if "platymatch" not in sys.modules:
        pkg = types.ModuleType("platymatch")
        pkg.__path__ = []
        sys.modules["platymatch"] = pkg
    for n in names:
        sys.modules.setdefault("platymatch." + n, importlib.import_module("platymatch_amd." + n))
    return sys.modules["platymatch"]


class HostArithmeticWarning(UserWarning):
    """self_check found that NumPy / BLAS on this host do not round the way the kernels restate them."""


_SELF_CHECK = {}
_SELF_CHECK_LOCK = __import__("threading").Lock()


def self_check(force=False):
    """Bit parity with the reference rests on three properties of the HOST's NumPy / BLAS that the kernels restate (DESIGN.md §4.4,
    §4.6, §5): np.add.reduce sums a long vector in pieces of np.getbufsize() elements with pairwise leaves (the centroid, the mean
    pairwise distance, the Similar-mode moments); BLAS ddot accumulates a 3-vector with fused multiply-adds (np.linalg.norm inside
    get_mean_distance, utils/utils.py:66); np.matmul of a 4 x 4 with a cloud is dgemm's fused chain (apply_affine_transform,
    apply_transform.py:13).  They hold for NumPy 1.2x / 2.x with OpenBLAS on x86-64; another BLAS or CPU may round differently, and
    then the REFERENCE's own numbers differ on that host — from the fixtures and from this package alike.  This compares the three
    on small inputs, host NumPy against the device kernels, once per process; a difference is reported as a HostArithmeticWarning
    (results stay correct to rounding; what is lost is the guarantee of the reference's exact bits on this machine).
    -> dict of the three verdicts.  Called by estimate_transform before its first registration; needs the GPU."""
    import warnings
    import numpy as np
    if _SELF_CHECK and not force:
        return dict(_SELF_CHECK)
    with _SELF_CHECK_LOCK:                    # (a batch's worker threads arrive together: one of them checks, the others wait)
        if _SELF_CHECK and not force:
            return dict(_SELF_CHECK)
        return _run_self_check()


def _run_self_check():
    import warnings
    import numpy as np
    from . import _kernels as K, _native as nat
    rng = np.random.default_rng(20241)
    x = np.ascontiguousarray(rng.normal(size=(3, 20011)) * 37.0 + 211.0)          # crosses two 8 192-element pieces and a ragged tail
    xd = nat.to_dev(x)
    ok = {}
    ok["np.mean: buffer pieces + pairwise leaves"] = bool(np.array_equal(K.centroid(xd).cpu().numpy(), x.mean(1)))
    small = np.ascontiguousarray(x[:, :45])
    d = [np.linalg.norm(small[:, i] - small[:, j]) for i in range(45) for j in range(i + 1, 45)]     # get_mean_distance's own list
    ok["BLAS ddot: fused multiply-adds"] = bool(K.mean_distance(nat.to_dev(small)).item() == np.average(d))
    A = np.eye(4)
    A[:3] = rng.normal(size=(3, 4))
    want = np.matmul(A, np.vstack((x[:, :4099], np.ones((1, 4099)))))[:3]
    got = K.apply_affine(nat.to_dev(A.reshape(16)), nat.to_dev(np.ascontiguousarray(x[:, :4099]))).cpu().numpy()
    ok["np.matmul: dgemm's fused chain"] = bool(np.array_equal(got, want))
    _SELF_CHECK.update(ok)
    bad = [k for k, v in ok.items() if not v]
    if bad:
        warnings.warn("platymatch_amd.self_check: this host's NumPy / BLAS differ from the arithmetic the kernels restate (%s; numpy %s, "
                      "np.getbufsize() = %d): results remain correct to rounding, but bit identity with the reference's output ON THIS "
                      "HOST is not guaranteed" % ("; ".join(bad), np.__version__, np.getbufsize()), HostArithmeticWarning, stacklevel=2)
    return dict(ok)
