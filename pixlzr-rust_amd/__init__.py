"""pixlzr-rust_amd — MI355X (gfx950) implementation of the pixlzr encode hot path.

The product is the C-ABI shared library csrc/libpixlzr_hip.so (include/pixlzr_hip.h):
hand-written HIP kernels + C++ host runtime.  This Python package is only the
ctypes plumbing tests and bench.py use to drive it (torch supplies device memory,
streams and torch.distributed).  There is no CPU fallback: if the library is
missing or no gfx950 device is present, calls fail loudly.
"""
from .binding import (  # noqa: F401
    FILTER_NEAREST, FILTER_TRIANGLE, FILTER_CATMULLROM, FILTER_GAUSSIAN, FILTER_LANCZOS3,
    MODE_SHRINK_BY, MODE_SHRINK_DIRECTIONALLY,
    DIST_OPAQUE, DIST_ALPHA, DIST_FLAT, DIST_NOISE,
    PxzError, Handle, build_library, library_path, load_library, grid, encode_container, qoi_encode, axis_table,
    EXPORTED_SYMBOLS,
)
from . import dist  # noqa: E402,F401  (torch.distributed plumbing: frame sharding + block-stream gather)
