"""Mirror of platymatch/estimate_transform/ (shape_context, find_transform, apply_transform, perform_icp)
plus the headless driver `estimate_transform(moving, fixed, ...)` (see platymatch_amd/pipeline.py)."""


def estimate_transform(moving, fixed, **kwargs):
    """Headless counterpart of the widget's worker (_dock_widget.py:526-718); see pipeline.estimate_transform."""
    from ..pipeline import estimate_transform as run
    return run(moving, fixed, **kwargs)
