"""ctypes binding of libdetexhip.so (include/detex.h + include/detexhip.h) plus torch plumbing.

torch is used only for what the task allows it for: device memory, streams and
torch.distributed.  Every decode goes through the C ABI into the HIP kernels; there is no
Python or CPU decode path here, and loading fails loudly if the built library is missing.
"""
import ctypes
import os

from . import formats as F

_HERE = os.path.dirname(os.path.abspath(__file__))
# DETEXHIP_LIB selects another build of the same ABI (the A/B measurement build, `make lib-ab`)
LIB_PATH = os.environ.get("DETEXHIP_LIB") or os.path.join(_HERE, "lib", "libdetexhip.so")
_lib = None

_vp = ctypes.c_void_p


class DetexHipError(RuntimeError):
    pass


def load():
    """Load detex_amd/lib/libdetexhip.so (built by `make lib` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        # torch's wheel bundles its own libamdhip64 (soname libamdhip64.so.7, requested by torch as
        # "libamdhip64.so"): it must be the first HIP runtime in the process, or torch would load a
        # second copy next to the /opt/rocm one libdetexhip.so pulls in and see no devices.
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise DetexHipError(
            "%s not found: build it with `make lib` (hipcc --offload-arch=gfx950). "
            "detex_amd has no fallback decode path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    lib.detexGetErrorMessage.restype = ctypes.c_char_p
    lib.detexhipVersion.restype = ctypes.c_char_p
    lib.detexhipKernelName.restype = ctypes.c_char_p
    lib.detexhipKernelName.argtypes = [ctypes.c_uint32]
    lib.detexhipSetDevice.argtypes = [ctypes.c_int]
    lib.detexhipSetKernelVariant.argtypes = [ctypes.c_int]
    lib.detexhipSetReadAhead.argtypes = [ctypes.c_int]
    lib.detexhipSetReadAhead.restype = ctypes.c_int
    lib.detexhipSetResidentIdleMicroseconds.argtypes = [ctypes.c_int]
    lib.detexhipSetResidentIdleMicroseconds.restype = ctypes.c_int
    lib.detexhipGetResidentStats.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(ctypes.c_ulonglong)]
    lib.detexhipGetResidentStats.restype = None
    lib.detexhipDecompressTextureLinearDevice.argtypes = [
        ctypes.c_uint32, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_size_t,
        ctypes.c_uint32, _vp, _vp]
    lib.detexhipDecompressTextureTiledDevice.argtypes = [
        ctypes.c_uint32, _vp, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_uint32, _vp, _vp]
    lib.detexhipDecompressBlocksDevice.argtypes = [
        ctypes.c_uint32, _vp, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_uint32, _vp, _vp, _vp]
    _lib = lib
    return lib


def last_error():
    m = load().detexGetErrorMessage()
    return None if m is None else m.decode()


def _check(rc, what):
    if rc != 0:
        raise DetexHipError("%s failed: %s" % (what, last_error()))


def set_kernel_variant(v):
    load().detexhipSetKernelVariant(int(v))


def set_read_ahead(mode):
    """detexhipSetReadAhead: 0 never, 1 (default; True) textures whose blocks exceed the Infinity Cache, 2 every texture with >= 1 MiB of blocks;
    returns the previous mode"""
    return load().detexhipSetReadAhead(int(mode))


def set_resident_idle_us(us):
    """idle time of the host tier's resident service kernel (0 = a launch per small call); returns the previous value"""
    return load().detexhipSetResidentIdleMicroseconds(int(us))


def resident_stats():
    """(requests answered by resident kernels, resident kernels started) of the calling thread"""
    a, b = ctypes.c_ulonglong(0), ctypes.c_ulonglong(0)
    load().detexhipGetResidentStats(ctypes.byref(a), ctypes.byref(b))
    return a.value, b.value


def kernel_name(fmt):
    n = load().detexhipKernelName(fmt.texture_format)
    return None if n is None else n.decode()


def _stream_handle(stream):
    import torch
    s = torch.cuda.current_stream() if stream is None else stream
    return ctypes.c_void_p(s.cuda_stream)


def decompress_linear_device(fmt, blocks, width, height, out=None, pitch=None, pixel_format=None,
                             status=None, stream=None, width_in_blocks=None, height_in_blocks=None):
    """detexhipDecompressTextureLinearDevice on torch CUDA tensors (uint8).  Asynchronous on the
    current (or given) torch stream.  Returns the output tensor (height, pitch) bytes."""
    import torch
    lib = load()
    pf = F.native_pixel_format(fmt) if pixel_format is None else pixel_format
    px = 1 + ((pf & 0xF00) >> 8)
    wb = (width + 3) // 4 if width_in_blocks is None else width_in_blocks
    hb = (height + 3) // 4 if height_in_blocks is None else height_in_blocks
    pitch = width * px if pitch is None else pitch
    assert blocks.is_cuda and blocks.dtype == torch.uint8 and blocks.is_contiguous()
    assert blocks.numel() >= wb * hb * fmt.block_bytes
    if out is None:
        out = torch.empty(height * pitch, dtype=torch.uint8, device=blocks.device)
    assert out.is_cuda and out.numel() >= (height - 1) * pitch + width * px if height > 0 else True
    _check(lib.detexhipDecompressTextureLinearDevice(
        fmt.texture_format, blocks.data_ptr(), width, height, wb, hb, out.data_ptr(), pitch, pf,
        _stream_handle(stream), None if status is None else status.data_ptr()),
        "detexhipDecompressTextureLinearDevice")
    return out


def decompress_tiled_device(fmt, blocks, wb, hb, out=None, pixel_format=None, status=None, stream=None):
    import torch
    lib = load()
    pf = F.native_pixel_format(fmt) if pixel_format is None else pixel_format
    px = 1 + ((pf & 0xF00) >> 8)
    if out is None:
        out = torch.empty(wb * hb * 16 * px, dtype=torch.uint8, device=blocks.device)
    _check(lib.detexhipDecompressTextureTiledDevice(
        fmt.texture_format, blocks.data_ptr(), wb, hb, out.data_ptr(), pf, _stream_handle(stream),
        None if status is None else status.data_ptr()), "detexhipDecompressTextureTiledDevice")
    return out


def decompress_blocks_device(fmt, blocks, n_blocks, mode_mask=F.MODE_MASK_ALL, flags=0, out=None, ok=None,
                             stream=None):
    """Batched per-block API: returns (pixels[n*16*px] uint8, ok[n] uint8)."""
    import torch
    lib = load()
    if out is None:
        out = torch.empty(n_blocks * 16 * fmt.pixel_bytes, dtype=torch.uint8, device=blocks.device)
    if ok is None:
        ok = torch.empty(max(n_blocks, 1), dtype=torch.uint8, device=blocks.device)
    _check(lib.detexhipDecompressBlocksDevice(
        fmt.texture_format, blocks.data_ptr(), n_blocks, mode_mask, flags, out.data_ptr(), ok.data_ptr(),
        _stream_handle(stream)), "detexhipDecompressBlocksDevice")
    return out, ok


def mode_histogram_device(fmt, blocks, n_blocks, hist=None, stream=None, accumulate=False):
    """detexhipModeHistogramDevice: 16 bins (uint32) of the reference's detexGetMode<FMT> values
    (accumulate=True: detexhipModeHistogramAccumulateDevice, added to `hist` instead of replacing it)."""
    import torch
    lib = load()
    if hist is None:
        hist = torch.zeros(16, dtype=torch.int32, device=blocks.device)
    name = "detexhipModeHistogramAccumulateDevice" if accumulate else "detexhipModeHistogramDevice"
    fn = getattr(lib, name)
    fn.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
    _check(fn(fmt.texture_format, blocks.data_ptr(), n_blocks, hist.data_ptr(), _stream_handle(stream)), name)
    return hist


class Shard(ctypes.Structure):
    """detexhipShard (include/detexhip.h)"""
    _fields_ = [("device", ctypes.c_int), ("d_blocks", ctypes.c_void_p), ("d_pixels", ctypes.c_void_p),
                ("row0", ctypes.c_int), ("row1", ctypes.c_int), ("decode_ms", ctypes.c_float), ("invalid_blocks", ctypes.c_int),
                ("peer_access", ctypes.c_int)]


def shard_rows(height_in_blocks, n_shards, shard):
    lib = load()
    r0, r1 = ctypes.c_int(), ctypes.c_int()
    _check(lib.detexhipShardRows(height_in_blocks, n_shards, shard, ctypes.byref(r0), ctypes.byref(r1)), "detexhipShardRows")
    return r0.value, r1.value


def decompress_linear_multi_device_host(fmt, host_blocks, width, height, devices, out=None, pitch=None, pixel_format=None,
                                        width_in_blocks=None, height_in_blocks=None):
    """detexhipDecompressTextureLinearMultiDeviceHost: numpy blocks in, numpy pixels out, one shard (and one PCIe link)
    per entry of `devices`.  Returns (ok, pixels, wall_ms)."""
    import numpy as np
    lib = load()
    lib.detexhipDecompressTextureLinearMultiDeviceHost.argtypes = [
        ctypes.c_uint32, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_size_t, ctypes.c_uint32,
        ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_float)]
    pf = F.native_pixel_format(fmt) if pixel_format is None else pixel_format
    px = 1 + ((pf & 0xF00) >> 8)
    wb = (width + 3) // 4 if width_in_blocks is None else width_in_blocks
    hb = (height + 3) // 4 if height_in_blocks is None else height_in_blocks
    pitch = width * px if pitch is None else pitch
    host_blocks = np.ascontiguousarray(host_blocks)
    if out is None:
        out = np.empty(height * pitch, np.uint8)
    devs = (ctypes.c_int * len(devices))(*devices)
    invalid, wall = ctypes.c_int(), ctypes.c_float()
    _check(lib.detexhipDecompressTextureLinearMultiDeviceHost(
        fmt.texture_format, host_blocks.ctypes.data_as(_vp), width, height, wb, hb, out.ctypes.data_as(_vp), pitch, pf, devs, len(devices),
        ctypes.byref(invalid), ctypes.byref(wall)), "detexhipDecompressTextureLinearMultiDeviceHost")
    return invalid.value == 0, out, wall.value


def decompress_linear_multi_device(fmt, width, height, devices, host_blocks=None, device_blocks=None, pixel_format=None,
                                   gather_device=-1, pitch=None):
    """detexhipDecompressTextureLinearMultiDevice: one texture, len(devices) shards of block rows, one calling
    thread.  host_blocks: numpy uint8 of the whole stream (uploaded by the call) or device_blocks: one torch uint8
    tensor per shard already on its device.  Returns dict(ok, bands=[torch tensors], gathered, shards, decode_wall_ms,
    gather_wall_ms)."""
    import numpy as np
    import torch
    lib = load()
    lib.detexhipDecompressTextureLinearMultiDevice.argtypes = [
        ctypes.c_uint32, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_uint32,
        ctypes.POINTER(Shard), ctypes.c_int, ctypes.c_int, _vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
    pf = F.native_pixel_format(fmt) if pixel_format is None else pixel_format
    px = 1 + ((pf & 0xF00) >> 8)
    wb, hb = (width + 3) // 4, (height + 3) // 4
    row_pitch = width * px if pitch is None else pitch
    n = len(devices)
    shards = (Shard * n)()
    bands = []
    for g, dev in enumerate(devices):
        r0, r1 = shard_rows(hb, n, g)
        rows = max(0, min(r1 * 4, height) - r0 * 4)
        band = torch.full((max(rows * row_pitch, 16),), 0xA5, dtype=torch.uint8, device="cuda:%d" % dev)
        bands.append(band)
        shards[g].device = dev
        shards[g].d_blocks = None if device_blocks is None else device_blocks[g].data_ptr()
        shards[g].d_pixels = band.data_ptr()
    gathered = None
    if gather_device >= 0:
        gathered = torch.full((height * row_pitch,), 0xA5, dtype=torch.uint8, device="cuda:%d" % gather_device)
    hb_ptr = None
    if host_blocks is not None:
        host_blocks = np.ascontiguousarray(host_blocks)
        hb_ptr = host_blocks.ctypes.data_as(ctypes.c_void_p)
    t_dec, t_gat = ctypes.c_float(), ctypes.c_float()
    torch.cuda.synchronize()
    _check(lib.detexhipDecompressTextureLinearMultiDevice(
        fmt.texture_format, hb_ptr, width, height, wb, hb, 0 if pitch is None else pitch, pf, shards, n, gather_device,
        None if gathered is None else gathered.data_ptr(), ctypes.byref(t_dec), ctypes.byref(t_gat)),
        "detexhipDecompressTextureLinearMultiDevice")
    return {"ok": all(s.invalid_blocks == 0 for s in shards), "bands": bands, "gathered": gathered,
            "shards": [(s.row0, s.row1, s.decode_ms, s.invalid_blocks) for s in shards], "peer_access": [s.peer_access for s in shards],
            "decode_wall_ms": t_dec.value, "gather_wall_ms": t_gat.value}
