#!/usr/bin/env python3
"""BC6H on coherent content (the fixture tiled: stream C) against uniform-random blocks (U): launch time of the linear kernel as a
function of the image's shape and row pitch, beside the block-major kernel -- what makes the linear layout slower on coherent content?
usage: python tools/gpu_bc6h_coherent.py [lib] [FORMAT=BPTC_FLOAT]"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from detex_amd import formats as F
import streams

vp = ctypes.c_void_p
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "detex_amd", "lib", "libdetexhip.so")
fmt = F.BY_NAME[sys.argv[2] if len(sys.argv) > 2 else "BPTC_FLOAT"]
px = fmt.pixel_bytes
lib = ctypes.CDLL(path)
lib.detexhipDecompressTextureLinearDevice.argtypes = [ctypes.c_uint32, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, ctypes.c_size_t, ctypes.c_uint32, vp, vp]
lib.detexhipDecompressTextureTiledDevice.argtypes = [ctypes.c_uint32, vp, ctypes.c_int, ctypes.c_int, vp, ctypes.c_uint32, vp, vp]
st = vp(torch.cuda.current_stream().cuda_stream)


def timed(step, n=150, rounds=3):
    for _ in range(400): step()
    best = None
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(n): step()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        best = us if best is None or us < best else best
    return round(best, 2)


for kind in ("C", "U"):
    for (W, H) in ((8192, 8192), (4096, 16384), (16384, 4096), (2048, 32768)):
        wb, hb = W // 4, H // 4
        data = streams.make_stream(kind, fmt, wb, hb)
        d = torch.from_numpy(np.ascontiguousarray(data)).cuda()
        row = {"format": fmt.name, "stream": kind, "W": W, "H": H}
        for pad in (0, 256, 4096, 4096 + 256, 65536 + 256):
            pitch = W * px + pad
            out = torch.empty(H * pitch + 256, dtype=torch.uint8, device="cuda")
            row["linear_pad%d" % pad] = timed(lambda: lib.detexhipDecompressTextureLinearDevice(fmt.texture_format, d.data_ptr(), W, H, wb, hb, out.data_ptr(), pitch, F.native_pixel_format(fmt), st, None))
            del out
        out = torch.empty(W * H * px, dtype=torch.uint8, device="cuda")
        row["tiled"] = timed(lambda: lib.detexhipDecompressTextureTiledDevice(fmt.texture_format, d.data_ptr(), wb, hb, out.data_ptr(), F.native_pixel_format(fmt), st, None))
        del out, d
        print(json.dumps(row), flush=True)
