"""SURVEY.md section 8(f) "next" rows: KTX1 loader (f-1), mip-chain batching (f-3), block-mode
histogram (f-4).  (f-2, the pixel-format epilogues, is covered in test_gpu_parity.py /
test_oracle_pin.py.)  CPU tests for the host-only parts, -m gpu tests for the kernels."""
import ctypes
import hashlib
import os
import struct

import numpy as np
import pytest

import oracle_lib as ol
import streams
from detex_amd import binding, formats as F
from detex_amd.ktx import read_ktx

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
_libc = ctypes.CDLL(None)
_libc.free.argtypes = [ctypes.c_void_p]
TexP = ctypes.POINTER(ol.DetexTexture)


def _loader(path_to_lib):
    lib = ctypes.CDLL(path_to_lib)
    lib.detexLoadKTXFile.argtypes = [ctypes.c_char_p, ctypes.POINTER(TexP)]
    lib.detexLoadKTXFile.restype = ctypes.c_bool
    lib.detexLoadKTXFileWithMipmaps.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.POINTER(TexP)), ctypes.POINTER(ctypes.c_int)]
    lib.detexLoadKTXFileWithMipmaps.restype = ctypes.c_bool
    lib.detexGetErrorMessage.restype = ctypes.c_char_p
    return lib


def _tex_fields(t):
    n = t.width_in_blocks * t.height_in_blocks * (8 + ((t.format & 0x00800000) >> 20))
    return (t.format, t.width, t.height, t.width_in_blocks, t.height_in_blocks, bytes(ctypes.string_at(t.data, n)))


def _free_tex(tp):
    _libc.free(ctypes.cast(tp.contents.data, ctypes.c_void_p))
    _libc.free(ctypes.cast(tp, ctypes.c_void_p))


def write_ktx(path, fmt, levels, big_endian=False, kv=b""):
    """levels: list of (width, height, data bytes)"""
    e = ">" if big_endian else "<"
    w, h, _ = levels[0]
    hdr = bytes([0xAB, 0x4B, 0x54, 0x58, 0x20, 0x31, 0x31, 0xBB, 0x0D, 0x0A, 0x1A, 0x0A])
    hdr += struct.pack(e + "13I", 0x04030201, 0, 1, 0, fmt.gl_internal_format, 0x1908, w, h, 0, 0, 1, len(levels), len(kv))
    body = kv
    for i, (_, _, data) in enumerate(levels):
        body += struct.pack(e + "I", len(data)) + bytes(data)
        if i + 1 < len(levels):
            body += b"\0" * (3 - ((len(data) + 3) % 4))
    open(path, "wb").write(hdr + body)


def mip_chain(fmt, width, height, seed):
    out = []
    w, h, k = width, height, 0
    while w >= 1 and h >= 1:
        wb, hb = (w + 3) // 4, (h + 3) // 4
        out.append((w, h, ol.stream_u(fmt, wb * hb, seed=seed + k).copy()))
        w >>= 1; h >>= 1; k += 1
    return out


# ---- f-1: KTX loader (host code only: runs without a GPU) ------------------------------------------
def test_ktx_loader_fixtures_match_python_reader_and_reference():
    ours = _loader(binding.LIB_PATH)
    ref = _loader(ol.REF_SO) if ol.have_ref() else None
    for f in F.FORMATS:
        if not f.fixture:
            continue
        path = os.path.join(GOLDEN, f.fixture).encode()
        tp = TexP()
        assert ours.detexLoadKTXFile(path, ctypes.byref(tp)), ours.detexGetErrorMessage()
        got = _tex_fields(tp.contents)
        k = read_ktx(path.decode())
        assert got == (f.texture_format, 64, 64, 16, 16, k["data"].tobytes())
        if ref is not None:
            rp = TexP()
            assert ref.detexLoadKTXFile(path, ctypes.byref(rp))
            assert _tex_fields(rp.contents) == got
            _free_tex(rp)
        _free_tex(tp)


@pytest.mark.parametrize("big_endian", [False, True])
def test_ktx_loader_mip_chain_and_metadata(tmp_path, big_endian):
    ours = _loader(binding.LIB_PATH)
    ref = _loader(ol.REF_SO) if ol.have_ref() else None
    for name in ("BC1", "BPTC", "EAC_R11"):
        fmt = F.BY_NAME[name]
        levels = mip_chain(fmt, 52, 20, seed=7)           # 52x20, 26x10, 13x5, 6x2, 3x1
        assert len(levels) == 5
        path = str(tmp_path / (name + ".ktx"))
        write_ktx(path, fmt, levels, big_endian=big_endian, kv=b"\x10\0\0\0KTXorient\0S=r\0\0\0" if not big_endian else b"")
        for lib in [l for l in (ours, ref) if l is not None]:
            arr = ctypes.POINTER(TexP)()
            n = ctypes.c_int()
            assert lib.detexLoadKTXFileWithMipmaps(path.encode(), 3, ctypes.byref(arr), ctypes.byref(n)), lib.detexGetErrorMessage()
            assert n.value == 3
            for i in range(3):
                w, h, data = levels[i]
                assert _tex_fields(arr[i].contents) == (fmt.texture_format, w, h, (w + 3) // 4, (h + 3) // 4, data.tobytes())
                _free_tex(arr[i])
            _libc.free(ctypes.cast(arr, ctypes.c_void_p))


def test_ktx_loader_errors(tmp_path):
    ours = _loader(binding.LIB_PATH)
    tp = TexP()
    assert not ours.detexLoadKTXFile(b"/nonexistent/x.ktx", ctypes.byref(tp))
    assert ours.detexGetErrorMessage() == b"detexLoadKTXFileWithMipmaps: Could not open file /nonexistent/x.ktx"
    bad = tmp_path / "bad.ktx"
    bad.write_bytes(b"\0" * 100)
    assert not ours.detexLoadKTXFile(str(bad).encode(), ctypes.byref(tp))
    assert ours.detexGetErrorMessage() == b"detexLoadKTXFileWithMipmaps: Couldn't find KTX signature"
    fmt = F.BY_NAME["BC1"]
    p = str(tmp_path / "size.ktx")
    write_ktx(p, fmt, [(8, 8, np.zeros(24, np.uint8))])          # should be 32 bytes
    assert not ours.detexLoadKTXFile(p.encode(), ctypes.byref(tp))
    assert b"does not match (24 vs 32)" in ours.detexGetErrorMessage()
    raw = bytearray(open(os.path.join(GOLDEN, "test-texture-BC1.ktx"), "rb").read())
    raw[28:32] = struct.pack("<I", 0x8058)                       # GL_RGBA8: an uncompressed payload
    q = tmp_path / "rgba8.ktx"
    q.write_bytes(bytes(raw))
    assert not ours.detexLoadKTXFile(str(q).encode(), ctypes.byref(tp))
    assert b"Unsupported format in .ktx file (glInternalFormat = 0x8058)" in ours.detexGetErrorMessage()
    # hostile headers (file content is untrusted): negative / huge dimensions, metadata length past the end of the file
    good = bytearray(open(os.path.join(GOLDEN, "test-texture-BC1.ktx"), "rb").read())
    for off, val in ((36, 0xFFFFFFF8), (36, 0), (36, 40000), (40, 0xFFFFFFF8), (40, 0x7FFFFFFF)):
        raw = bytearray(good)
        raw[off:off + 4] = struct.pack("<I", val)
        q = tmp_path / ("dim_%d_%x.ktx" % (off, val))
        q.write_bytes(bytes(raw))
        assert not ours.detexLoadKTXFile(str(q).encode(), ctypes.byref(tp)), (off, hex(val))
        assert b"is outside 1..32768" in ours.detexGetErrorMessage(), ours.detexGetErrorMessage()
    raw = bytearray(good)
    raw[60:64] = struct.pack("<I", 1 << 30)                      # bytesOfKeyValueData far beyond the file
    q = tmp_path / "kv.ktx"
    q.write_bytes(bytes(raw))
    assert not ours.detexLoadKTXFile(str(q).encode(), ctypes.byref(tp))
    assert b"Error reading file" in ours.detexGetErrorMessage()
    q = tmp_path / "short.ktx"
    q.write_bytes(bytes(good[:64 + 4 + 100]))                    # payload truncated
    assert not ours.detexLoadKTXFile(str(q).encode(), ctypes.byref(tp))
    assert b"Error reading file" in ours.detexGetErrorMessage()


@pytest.mark.skipif(not ol.have_ref(), reason="needs oracle/_ref")
def test_mode_classifier_matches_reference_getmode(oracle):
    """pins oracle.modes (what the histogram kernel is checked against) to the reference's detexGetMode<FMT>"""
    ref = ctypes.CDLL(ol.REF_SO)
    for name in ("BC1", "BPTC", "BPTC_FLOAT", "BPTC_SIGNED_FLOAT", "ETC1", "ETC2", "ETC2_PUNCHTHROUGH", "ETC2_EAC"):   # BC1A has no detexGetMode of its own
        fmt = F.BY_NAME[name]
        fn = getattr(ref, "detexGetMode" + name)
        fn.argtypes = [ctypes.POINTER(ctypes.c_uint8)]
        fn.restype = ctypes.c_int32
        blocks, _ = streams.forced_stream(fmt)
        want = np.array([fn(ol._ptr(b)) for b in blocks], np.int32)
        assert np.array_equal(oracle.modes(fmt, blocks), want), name


# ---- GPU ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.mark.gpu
def test_reference_call_sequence_file_to_pixels(hiplib, golden_json):
    """validate.c:135,199-209 end to end against libdetexhip for ALL 17 bundled fixtures: detexLoadKTXFile ->
    detexDecompressTextureLinear(BGRA8 if detexFormatHasAlpha(format) else BGRX8) -- the call the reference's own
    validation program makes for every format, including the one-/two-component, signed and half-float ones"""
    ours = _loader(binding.LIB_PATH)
    want = golden_json("fixtures.json")
    for f in F.FORMATS:
        if not f.fixture:
            continue
        tp = TexP()
        assert ours.detexLoadKTXFile(os.path.join(GOLDEN, f.fixture).encode(), ctypes.byref(tp))
        pf = F.PIXEL_FORMAT_BGRA8 if (f.texture_format & 0x4) else F.PIXEL_FORMAT_BGRX8       # detexFormatHasAlpha, detex.h:89,907-909
        assert pf in F.accepted_pixel_formats(f), f.name
        out = np.zeros(64 * 64 * 4, np.uint8)
        ok = hiplib.lib.detexDecompressTextureLinear(tp, ol._ptr(out), pf)
        g = want[f.name]["0x%04X" % pf]
        assert bool(ok) == g["ok"] and sha(out) == g["sha256"], f.name
        _free_tex(tp)


@pytest.mark.gpu
@pytest.mark.parametrize("name,pf", [("BC1", None), ("BC3", F.PIXEL_FORMAT_RGB8), ("BPTC", F.PIXEL_FORMAT_BGRA8), ("BPTC_FLOAT", None),
                                     ("ETC2_EAC", None), ("RGTC1", None), ("EAC_SIGNED_RG11", None)])
def test_mip_chain_one_launch(name, pf, torch_cuda, hiplib, oracle):
    torch = torch_cuda
    fmt = F.BY_NAME[name]
    pf = F.native_pixel_format(fmt) if pf is None else pf
    tpx = 1 + ((pf & 0xF00) >> 8)
    levels = mip_chain(fmt, 1024, 512, seed=0x3117 + fmt.index)      # 1024x512 ... 2x1 : 10 levels
    assert len(levels) == 10
    lib = binding.load()

    class Level(ctypes.Structure):
        _fields_ = [("d_blocks", ctypes.c_void_p), ("d_pixels", ctypes.c_void_p), ("pitch", ctypes.c_size_t),
                    ("width", ctypes.c_int), ("height", ctypes.c_int), ("wb", ctypes.c_int), ("hb", ctypes.c_int)]
    d_in = [torch.from_numpy(d).cuda() for _, _, d in levels]
    d_out = [torch.full((w * h * tpx + 64,), 0xA5, dtype=torch.uint8, device="cuda") for w, h, _ in levels]
    arr = (Level * len(levels))(*[Level(d_in[i].data_ptr(), d_out[i].data_ptr(), w * tpx, w, h, (w + 3) // 4, (h + 3) // 4)
                                  for i, (w, h, _) in enumerate(levels)])
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    lib.detexhipDecompressLevelsLinearDevice.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
    rc = lib.detexhipDecompressLevelsLinearDevice(fmt.texture_format, arr, len(levels), pf, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), status.data_ptr())
    assert rc == 0, binding.last_error()
    torch.cuda.synchronize()
    all_ok = True
    for i, (w, h, data) in enumerate(levels):
        ok_o, want = oracle.linear_to(fmt, data, w, h, pf)
        all_ok &= ok_o
        got = d_out[i].cpu().numpy()
        assert np.array_equal(got[:w * h * tpx], want), (name, i, w, h)
        assert (got[w * h * tpx:] == 0xA5).all()
    assert bool(status.item() == 0) == all_ok
    # a level whose blocks are not aligned to the block size is refused like the one-texture entry refuses it (a block is ONE 8 / 16-byte load)
    arr[1].d_blocks = d_in[1].data_ptr() + 4
    assert lib.detexhipDecompressLevelsLinearDevice(fmt.texture_format, arr, len(levels), pf, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), status.data_ptr()) == 1
    assert "level 1" in binding.last_error() and "aligned" in binding.last_error()
    # host tier: the same chain through detexTexture structs, one call
    texs = [ol.DetexTexture(fmt.texture_format, ol._ptr(d), w, h, (w + 3) // 4, (h + 3) // 4) for w, h, d in levels]
    outs = [np.zeros(w * h * tpx, np.uint8) for w, h, _ in levels]
    tarr = (ctypes.POINTER(ol.DetexTexture) * len(levels))(*[ctypes.pointer(t) for t in texs])
    oarr = (ctypes.POINTER(ctypes.c_uint8) * len(levels))(*[ol._ptr(o) for o in outs])
    lib.detexhipDecompressTexturesLinear.restype = ctypes.c_bool
    r = lib.detexhipDecompressTexturesLinear(tarr, len(levels), oarr, ctypes.c_uint32(pf))
    assert bool(r) == all_ok
    for i, (w, h, data) in enumerate(levels):
        assert np.array_equal(outs[i], oracle.linear_to(fmt, data, w, h, pf)[1]), (name, i)


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", F.FORMATS, ids=[f.name for f in F.FORMATS])
def test_mode_histogram(fmt, torch_cuda, oracle):
    torch = torch_cuda
    lib = binding.load()
    blocks = np.concatenate([streams.forced_stream(fmt)[0], ol.stream_u(fmt, 300001, seed=11).reshape(-1, fmt.block_bytes)])
    modes = oracle.modes(fmt, blocks)
    want = np.bincount(np.where(modes < 0, 15, modes), minlength=16).astype(np.uint32)
    hist = np.zeros(16, np.uint32)
    lib.detexhipModeHistogram.restype = ctypes.c_bool
    lib.detexhipModeHistogram.argtypes = [ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint8), ctypes.c_size_t, ctypes.c_void_p]
    assert lib.detexhipModeHistogram(fmt.texture_format, ol._ptr(np.ascontiguousarray(blocks)), len(blocks), hist.ctypes.data)
    assert np.array_equal(hist, want), (fmt.name, hist, want)
    assert hist.sum() == len(blocks)
    # the accumulating device entry: two parts of the stream added into one histogram that starts non-zero
    d = torch.from_numpy(np.ascontiguousarray(blocks).reshape(-1)).cuda()
    split = (len(blocks) // 3) * fmt.block_bytes
    start = torch.arange(16, dtype=torch.int32, device="cuda")
    acc = start.clone()
    binding.mode_histogram_device(fmt, d[:split], split // fmt.block_bytes, hist=acc, accumulate=True)
    binding.mode_histogram_device(fmt, d[split:], len(blocks) - split // fmt.block_bytes, hist=acc, accumulate=True)
    torch.cuda.synchronize()
    assert np.array_equal(acc.cpu().numpy().astype(np.uint32), want + np.arange(16, dtype=np.uint32)), fmt.name
