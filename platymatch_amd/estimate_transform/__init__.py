"""Mirror of platymatch/estimate_transform/: shape_context, find_transform, apply_transform, perform_icp."""
