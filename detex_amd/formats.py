"""Format table of the detex block-decode path (interface facts of detex.h).

Values mirror the reference's public enums so that callers can pass the same constants:
compressed-format index = ``texture_format >> 24`` (detex.h:577-611, texture.c:27-48),
``block_bytes = 8 + ((fmt & 0x00800000) >> 20)`` (detex.h:918-920),
``pixel_bytes = 1 + ((fmt & 0xF00) >> 8)`` (detex.h:879-881).
"""
from collections import namedtuple

# pixel formats (detex.h:83-379)
PIXEL_FORMAT_RGBA8 = 0x334
PIXEL_FORMAT_RGBX8 = 0x320
PIXEL_FORMAT_R8 = 0x000
PIXEL_FORMAT_RG8 = 0x110
PIXEL_FORMAT_R16 = 0x101
PIXEL_FORMAT_SIGNED_R16 = 0x1101
PIXEL_FORMAT_RG16 = 0x311
PIXEL_FORMAT_SIGNED_RG16 = 0x1311
PIXEL_FORMAT_FLOAT_RGBX16 = 0x2721
PIXEL_FORMAT_SIGNED_FLOAT_RGBX16 = 0x3721

MODE_MASK_ALL = 0xFFFFFFFF
FLAG_ENCODE = 0x1
FLAG_OPAQUE_ONLY = 0x2
FLAG_NON_OPAQUE_ONLY = 0x4

TextureFormat = namedtuple(
    "TextureFormat", "name index texture_format block_bytes pixel_bytes gl_internal_format fixture")

# name, texture_format (detex.h:613-727), glInternalFormat of the KTX fixture (file-info.c:75-97)
_ROWS = [
    ("BC1", 0x01000320, 0x83F0, True),
    ("BC1A", 0x02000334, 0x83F1, True),
    ("BC2", 0x03800334, 0x83F2, True),
    ("BC3", 0x04800334, 0x83F3, True),
    ("RGTC1", 0x05000000, 0x8DBB, True),
    ("SIGNED_RGTC1", 0x06001101, 0x8DBC, True),
    ("RGTC2", 0x07800110, 0x8DBD, True),
    ("SIGNED_RGTC2", 0x08801311, 0x8DBE, True),
    ("BPTC_FLOAT", 0x09802721, 0x8E8F, True),
    ("BPTC_SIGNED_FLOAT", 0x0A803721, 0x8E8E, False),   # no fixture in the reference tree
    ("BPTC", 0x0B800334, 0x8E8C, True),
    ("ETC1", 0x0C000320, 0x8D64, True),
    ("ETC2", 0x0D000320, 0x9274, True),
    ("ETC2_PUNCHTHROUGH", 0x0E000334, 0x9275, True),
    ("ETC2_EAC", 0x0F800334, 0x9278, True),
    ("EAC_R11", 0x10000101, 0x9270, True),
    ("EAC_SIGNED_R11", 0x11001101, 0x9271, True),
    ("EAC_RG11", 0x12800311, 0x9272, True),
    ("EAC_SIGNED_RG11", 0x13801311, 0x9273, False),     # no fixture in the reference tree
]


def _mk(name, tf, gl, has_fixture):
    return TextureFormat(name, tf >> 24, tf, 8 + ((tf & 0x00800000) >> 20), 1 + ((tf & 0xF00) >> 8), gl,
                         ("test-texture-%s.ktx" % name) if has_fixture else None)


FORMATS = [_mk(*r) for r in _ROWS]
BY_NAME = {f.name: f for f in FORMATS}
BY_INDEX = {f.index: f for f in FORMATS}
BY_GL = {f.gl_internal_format: f for f in FORMATS}


def native_pixel_format(fmt):
    return fmt.texture_format & 0xFFFF


_PIXEL_NAMES = {0x334: "RGBA8", 0x320: "RGBA8", 0x000: "R8", 0x110: "RG8", 0x101: "R16", 0x1101: "SIGNED_R16",
                0x311: "RG16", 0x1311: "SIGNED_RG16", 0x2721: "FLOAT_RGBX16", 0x3721: "SIGNED_FLOAT_RGBX16"}


def target_name(fmt):
    """display name of the decode target (RGBX8 natives are byte-identical to RGBA8, alpha = 0xFF)"""
    return _PIXEL_NAMES[native_pixel_format(fmt)]


PIXEL_FORMAT_BGRA8 = 0x33C
PIXEL_FORMAT_BGRX8 = 0x328
PIXEL_FORMAT_RGB8 = 0x220
PIXEL_FORMAT_FLOAT_BGRX16 = 0x2729
PIXEL_FORMAT_NAMES = {**_PIXEL_NAMES, 0x33C: "BGRA8", 0x328: "BGRX8", 0x220: "RGB8", 0x2729: "FLOAT_BGRX16"}

# epilogue kinds (detex_amd/csrc/kernels.h, oracle orc_convert_pixels): 0 none, 1 swap R/B (8-bit), 2 pack RGB8,
# 3 swap R/B (16-bit), 4/5/6 one-/two-component and half-float natives -> RGBX8 / BGRX8 / RGB8
_SMALL_UNSIGNED = (PIXEL_FORMAT_R8, PIXEL_FORMAT_RG8, PIXEL_FORMAT_R16, PIXEL_FORMAT_RG16, PIXEL_FORMAT_FLOAT_RGBX16)
_SMALL_SIGNED = (PIXEL_FORMAT_SIGNED_R16, PIXEL_FORMAT_SIGNED_RG16)


def epilogue_kind(fmt, pixel_format):
    n = native_pixel_format(fmt)
    if pixel_format == n:
        return 0
    if n in (PIXEL_FORMAT_RGBA8, PIXEL_FORMAT_RGBX8):
        return {PIXEL_FORMAT_RGBA8: 0, PIXEL_FORMAT_RGBX8: 0, PIXEL_FORMAT_BGRA8: 1, PIXEL_FORMAT_BGRX8: 1,
                PIXEL_FORMAT_RGB8: 2}.get(pixel_format)
    if n == PIXEL_FORMAT_FLOAT_RGBX16 and pixel_format == PIXEL_FORMAT_FLOAT_BGRX16:
        return 3
    if n in _SMALL_UNSIGNED or n in _SMALL_SIGNED:
        if n in _SMALL_SIGNED and pixel_format == PIXEL_FORMAT_BGRA8:
            return None                       # the reference finds no conversion path either (convert.c:885-1063)
        return {PIXEL_FORMAT_RGBA8: 4, PIXEL_FORMAT_RGBX8: 4, PIXEL_FORMAT_BGRA8: 5, PIXEL_FORMAT_BGRX8: 5,
                PIXEL_FORMAT_RGB8: 6}.get(pixel_format)
    return None


ALL_TARGETS = (PIXEL_FORMAT_RGBA8, PIXEL_FORMAT_RGBX8, PIXEL_FORMAT_BGRA8, PIXEL_FORMAT_BGRX8, PIXEL_FORMAT_RGB8,
               PIXEL_FORMAT_FLOAT_BGRX16)


def accepted_pixel_formats(fmt):
    """Target pixel formats the block-decode path supports for ``fmt``: the native one, the RGBX8<->RGBA8 no-op
    edge of the reference's conversion table (convert.c:768-769), and the in-kernel epilogues: every 8-bit RGB(A)
    target the reference's callers request (BGRA8/BGRX8, RGB8, RGBA8/RGBX8) for every format the reference itself
    can convert, FLOAT_BGRX16 for unsigned BC6H."""
    n = native_pixel_format(fmt)
    out = [n] + [pf for pf in ALL_TARGETS if pf != n and epilogue_kind(fmt, pf) is not None]
    return tuple(out)


def target_pixel_bytes(pixel_format):
    return 1 + ((pixel_format & 0xF00) >> 8)
