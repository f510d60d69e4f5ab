"""Minimal KTX1 reader for the reference's bundled ``test-texture-*.ktx`` fixtures.

Covers exactly what those files use (the reference's loader is
ktx.c:36-176): little-endian KTX 1.1 header (64 bytes, 12-byte identifier + 13 u32 words),
``bytesOfKeyValueData`` skipped, then per mip level a u32 ``imageSize`` followed by the payload.
Only the first mip level is returned.  Not a general KTX library (containers are out of scope).
"""
import struct

import numpy as np

from . import formats as F

_KTX_ID = bytes([0xAB, 0x4B, 0x54, 0x58, 0x20, 0x31, 0x31, 0xBB, 0x0D, 0x0A, 0x1A, 0x0A])


def read_ktx(path):
    raw = open(path, "rb").read()
    if raw[:12] != _KTX_ID:
        raise ValueError("%s: not a KTX1 file" % path)
    (endian, gl_type, gl_type_size, gl_format, gl_internal, gl_base, width, height, depth, n_array,
     n_faces, n_mips, kv_bytes) = struct.unpack_from("<13I", raw, 12)
    if endian != 0x04030201:
        raise ValueError("%s: big-endian KTX not supported" % path)
    fmt = F.BY_GL.get(gl_internal)
    if fmt is None:
        raise ValueError("%s: glInternalFormat 0x%04X is not a block format of this path" % (path, gl_internal))
    off = 64 + kv_bytes
    (image_size,) = struct.unpack_from("<I", raw, off)
    height = max(height, 1)
    wb, hb = (width + 3) // 4, (height + 3) // 4
    need = wb * hb * fmt.block_bytes
    if image_size < need or len(raw) < off + 4 + need:
        raise ValueError("%s: truncated payload" % path)
    data = np.frombuffer(raw, np.uint8, need, off + 4).copy()
    return {"format": fmt, "width": width, "height": height, "width_in_blocks": wb,
            "height_in_blocks": hb, "data": data, "mip_levels": n_mips}
