"""deepcharuco_amd -- MI355X-native Deep ChArUco detect+refine path.

Drop-in for the reference's ``src/inference.py`` API (``load_models`` / ``infer_image`` /
``solve_pnp``) backed by hand-written gfx950 HIP kernels behind a C ABI
(``include/deepcharuco_amd.h`` -> ``libdeepcharuco_amd.so``).  Importing this package does
not load the library; the first call into it does, and fails loudly if it is not built.
"""
__version__ = "0.1.0"
