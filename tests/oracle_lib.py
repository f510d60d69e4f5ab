"""Test-side loaders for the CPU checkers (oracle restatement and, when present, the compiled
reference).  Test infrastructure only -- nothing in the product imports this."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libdetex_ref.so")
_u8p = ctypes.POINTER(ctypes.c_uint8)


def _ptr(a):
    return a.ctypes.data_as(_u8p)


class Oracle:
    """oracle/libdetex_oracle.so (our plain-C restatement)."""

    def __init__(self, sanitized=False):
        name = "libdetex_oracle_san.so" if sanitized else "libdetex_oracle.so"
        path = os.path.join(ORACLE_DIR, name)
        src = os.path.join(ORACLE_DIR, "detex_oracle.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, name])
        self.lib = lib = ctypes.CDLL(path)
        lib.orc_decode_block.argtypes = [ctypes.c_int, _u8p, ctypes.c_uint32, ctypes.c_uint32, _u8p]
        lib.orc_decompress_linear.argtypes = [ctypes.c_int, _u8p] + [ctypes.c_int] * 4 + [_u8p]
        lib.orc_decompress_tiled.argtypes = [ctypes.c_int, _u8p, ctypes.c_int, ctypes.c_int, _u8p]
        lib.orc_block_mode.argtypes = [ctypes.c_int, _u8p]
        lib.orc_block_modes.argtypes = [ctypes.c_int, _u8p, ctypes.c_long, ctypes.c_void_p]
        lib.orc_decode_blocks.argtypes = [ctypes.c_int, _u8p, ctypes.c_long, ctypes.c_uint32, ctypes.c_uint32,
                                          _u8p, _u8p]

    def block(self, fmt, data, mode_mask=0xFFFFFFFF, flags=0):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        out = np.zeros(16 * fmt.pixel_bytes, np.uint8)
        ok = self.lib.orc_decode_block(fmt.index, _ptr(data), mode_mask, flags, _ptr(out))
        return bool(ok), out

    def linear(self, fmt, data, width, height, wb=None, hb=None):
        """wb, hb: the texture's block grid where it is not the image's (texture.c:105-145 loops over the grid and clips to the image)"""
        wb = (width + 3) // 4 if wb is None else wb
        hb = (height + 3) // 4 if hb is None else hb
        data = np.ascontiguousarray(data, dtype=np.uint8)
        assert data.size >= wb * hb * fmt.block_bytes
        out = np.zeros(width * height * fmt.pixel_bytes, np.uint8)
        ok = self.lib.orc_decompress_linear(fmt.index, _ptr(data), width, height, wb, hb, _ptr(out))
        return bool(ok), out

    def tiled(self, fmt, data, wb, hb):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        out = np.zeros(wb * hb * 16 * fmt.pixel_bytes, np.uint8)
        ok = self.lib.orc_decompress_tiled(fmt.index, _ptr(data), wb, hb, _ptr(out))
        return bool(ok), out

    def convert(self, fmt, native_pixels, pixel_format):
        """native pixels -> one of the target formats the GPU path offers as an epilogue"""
        from detex_amd import formats as F
        kind = F.epilogue_kind(fmt, pixel_format)
        assert kind is not None, "target 0x%X not offered for %s" % (pixel_format, fmt.name)
        native_pixels = np.ascontiguousarray(native_pixels, np.uint8).reshape(-1)
        if kind == 0:
            return native_pixels
        n = native_pixels.size // fmt.pixel_bytes
        out = np.zeros(n * F.target_pixel_bytes(pixel_format), np.uint8)
        self.lib.orc_convert_pixels.restype = ctypes.c_long
        self.lib.orc_convert_pixels.argtypes = [ctypes.c_uint32, ctypes.c_int, _u8p, ctypes.c_long, _u8p]
        assert self.lib.orc_convert_pixels(F.native_pixel_format(fmt), kind, _ptr(native_pixels), n, _ptr(out)) == out.size
        return out

    def _zero_failed_blocks(self, fmt, data, converted, tpx, wb, hb, width=None, height=None):
        """texture.c:125-128: a block whose decode fails is zero-filled in the TARGET format (the conversion never runs
        for it), so its pixels are zero bytes -- not converted zeros (X / alpha = 0xFF)"""
        okb, _ = self.blocks(fmt, np.ascontiguousarray(data, np.uint8)[:wb * hb * fmt.block_bytes].reshape(-1, fmt.block_bytes))
        bad = np.flatnonzero(~okb)
        if bad.size == 0:
            return converted
        if width is None:                                   # block-major layout
            v = converted.reshape(-1, 16 * tpx)
            v[bad] = 0
            return converted
        img = converted.reshape(height, width * tpx)
        for b in bad:
            by, bx = divmod(int(b), wb)
            img[by * 4:by * 4 + 4, bx * 4 * tpx:(bx * 4 + 4) * tpx] = 0
        return converted

    def linear_to(self, fmt, data, width, height, pixel_format):
        from detex_amd import formats as F
        ok, out = self.linear(fmt, data, width, height)
        conv = self.convert(fmt, out, pixel_format)
        if not ok and F.epilogue_kind(fmt, pixel_format) >= 4:
            conv = self._zero_failed_blocks(fmt, data, conv.copy(), F.target_pixel_bytes(pixel_format), (width + 3) // 4, (height + 3) // 4, width, height)
        return ok, conv

    def tiled_to(self, fmt, data, wb, hb, pixel_format):
        from detex_amd import formats as F
        ok, out = self.tiled(fmt, data, wb, hb)
        conv = self.convert(fmt, out, pixel_format)
        if not ok and F.epilogue_kind(fmt, pixel_format) >= 4:
            conv = self._zero_failed_blocks(fmt, data, conv.copy(), F.target_pixel_bytes(pixel_format), wb, hb)
        return ok, conv

    def modes(self, fmt, data):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        n = data.size // fmt.block_bytes
        out = np.zeros(n, np.int32)
        self.lib.orc_block_modes(fmt.index, _ptr(data), n, out.ctypes.data_as(ctypes.c_void_p))
        return out

    def blocks(self, fmt, data, mode_mask=0xFFFFFFFF, flags=0):
        """Per-block decode of n blocks -> (ok[n], pixels[n, 16*px]); failed blocks zeroed here."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        n = data.size // fmt.block_bytes
        out = np.zeros((n, 16 * fmt.pixel_bytes), np.uint8)
        ok = np.zeros(n, np.uint8)
        self.lib.orc_decode_blocks(fmt.index, _ptr(data), n, mode_mask, flags, _ptr(out), _ptr(ok))
        out[ok == 0] = 0
        return ok.astype(bool), out


class DetexTexture(ctypes.Structure):
    """detexTexture, detex.h:729-736 (sizeof 32 on LP64)."""
    _fields_ = [("format", ctypes.c_uint32), ("data", _u8p), ("width", ctypes.c_int),
                ("height", ctypes.c_int), ("width_in_blocks", ctypes.c_int),
                ("height_in_blocks", ctypes.c_int)]


class DetexAPI:
    """ctypes binding of the detex C API (detex.h:435-531, 747-765, 806) over any library that
    exports it: the compiled reference (oracle/_ref) or our libdetexhip.so."""

    def __init__(self, path):
        self.path = path
        self.lib = lib = ctypes.CDLL(path)
        for n in ("detexDecompressTextureLinear", "detexDecompressTextureTiled"):
            f = getattr(lib, n)
            f.argtypes = [ctypes.POINTER(DetexTexture), _u8p, ctypes.c_uint32]
            f.restype = ctypes.c_bool
        lib.detexDecompressBlock.argtypes = [_u8p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                             _u8p, ctypes.c_uint32]
        lib.detexDecompressBlock.restype = ctypes.c_bool
        lib.detexGetErrorMessage.restype = ctypes.c_char_p

    def block_fn(self, fmt):
        f = getattr(self.lib, "detexDecompressBlock" + fmt.name)
        f.argtypes = [_u8p, ctypes.c_uint32, ctypes.c_uint32, _u8p]
        f.restype = ctypes.c_bool
        return f

    def _texture(self, fmt, data, width, height, wb=None, hb=None):
        wb = (width + 3) // 4 if wb is None else wb
        hb = (height + 3) // 4 if hb is None else hb
        return DetexTexture(fmt.texture_format, _ptr(data), width, height, wb, hb)

    def linear(self, fmt, data, width, height, pixel_format=None, out=None, wb=None, hb=None):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        pf = (fmt.texture_format & 0xFFFF) if pixel_format is None else pixel_format
        px = 1 + ((pf & 0xF00) >> 8)
        if out is None:
            out = np.zeros(width * height * px, np.uint8)
        tex = self._texture(fmt, data, width, height, wb, hb)
        ok = self.lib.detexDecompressTextureLinear(ctypes.byref(tex), _ptr(out), pf)
        return bool(ok), out

    def tiled(self, fmt, data, wb, hb, pixel_format=None):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        pf = (fmt.texture_format & 0xFFFF) if pixel_format is None else pixel_format
        px = 1 + ((pf & 0xF00) >> 8)
        out = np.zeros(wb * hb * 16 * px, np.uint8)
        tex = self._texture(fmt, data, wb * 4, hb * 4, wb, hb)
        ok = self.lib.detexDecompressTextureTiled(ctypes.byref(tex), _ptr(out), pf)
        return bool(ok), out

    def block(self, fmt, data, mode_mask=0xFFFFFFFF, flags=0):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        out = np.zeros(16 * fmt.pixel_bytes, np.uint8)
        ok = self.block_fn(fmt)(_ptr(data), mode_mask, flags, _ptr(out))
        return bool(ok), out

    def blocks(self, fmt, data, mode_mask=0xFFFFFFFF, flags=0, want_ok=True):
        """detexhipDecompressBlocks (libdetexhip only; include/detexhip.h): n blocks in one call -> (all_ok, ok[n] or None, pixels[n, 16*px])"""
        f = self.lib.detexhipDecompressBlocks
        f.argtypes = [ctypes.c_uint32, _u8p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_uint32, _u8p, _u8p]
        f.restype = ctypes.c_bool
        data = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1)
        n = data.size // fmt.block_bytes
        out = np.full((n, 16 * fmt.pixel_bytes), 0xA5, np.uint8)
        ok = np.full(n, 0xA5, np.uint8) if want_ok else None
        r = f(fmt.texture_format, _ptr(data), n, mode_mask, flags, _ptr(out), _ptr(ok) if want_ok else None)
        return bool(r), ok, out

    def error(self):
        m = self.lib.detexGetErrorMessage()
        return None if m is None else m.decode()


def have_ref():
    return os.path.exists(REF_SO)


def load_ref():
    return DetexAPI(REF_SO)


# ---------------------------------------------------------------------------------------------
# synthetic block streams
# ---------------------------------------------------------------------------------------------
STREAM_SEED_BASE = 0xD37E5000
STREAM_SEED_K = {"BC1": 0, "BC3": 1, "BPTC": 2, "ETC2": 3, "ETC2_EAC": 4, "BPTC_FLOAT": 5}


def splitmix64_words(seed, n, first=0):
    """words [first, first + n) of the splitmix64 stream as little-endian u64 (vectorised, identical to the scalar recurrence:
    word k depends on k only, so any slice of a stream is computed without the words before it)."""
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * np.arange(first + 1, first + n + 1, dtype=np.uint64))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def stream_u(fmt, n_blocks, seed=None):
    """Stream U: uniform random bytes from splitmix64; default seed is the survey's per-format one."""
    if seed is None:
        seed = STREAM_SEED_BASE + STREAM_SEED_K.get(fmt.name, 16 + fmt.index)
    return splitmix64_words(seed, n_blocks * fmt.block_bytes // 8).view(np.uint8)


def stream_u_slice(fmt, first_block, n_blocks, seed=None):
    """blocks [first_block, first_block + n_blocks) of stream U: what a rank that owns a band of a sharded image materialises
    (detex_amd/sharding.py: shard_of(...).in_offset / in_bytes), identical to the same slice of stream_u(fmt, total)"""
    if seed is None:
        seed = STREAM_SEED_BASE + STREAM_SEED_K.get(fmt.name, 16 + fmt.index)
    wpb = fmt.block_bytes // 8
    return splitmix64_words(seed, n_blocks * wpb, first_block * wpb).view(np.uint8)


def fnv1a64(buf):
    h = 0xcbf29ce484222325
    for b in memoryview(np.ascontiguousarray(buf, np.uint8)).tobytes():
        h = ((h ^ b) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h
