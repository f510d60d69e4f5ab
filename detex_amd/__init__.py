"""detex_amd -- MI355X-native texture-block decompression behind the detex C API.

The product is the C-ABI shared library ``detex_amd/lib/libdetexhip.so`` (hand-written HIP
kernels for gfx950 + a C++ host shim exporting the reference's ``detexDecompress*`` entry
points, see include/detex.h and include/detexhip.h).  This Python package is only the thin
host-side mirror used by the tests, bench.py and the multi-GPU launcher: format constants, a
KTX1 fixture reader, a ctypes binding of the C ABI and torch plumbing for device buffers.
"""
from . import formats  # noqa: F401
