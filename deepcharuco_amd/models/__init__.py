"""Mirrors of the reference's ``src/models`` package (net, refinenet, model_utils), HIP-backed."""
