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
