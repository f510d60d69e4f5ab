#!/usr/bin/env python3
"""Launch time of the linear decode as a function of the image's row pitch / width (which kernel the dispatch picks, and what
row alignment does to it).  usage: python tools/gpu_pitch_sweep.py lib1[,lib2...] [FORMAT=BC1]"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from detex_amd import formats as F
import oracle_lib as ol

vp = ctypes.c_void_p
fmt = F.BY_NAME[sys.argv[2] if len(sys.argv) > 2 else "BC1"]
px = fmt.pixel_bytes
st = vp(torch.cuda.current_stream().cuda_stream)
cases = [(8192, 8192, 0), (8192, 8192, 64), (8192, 8192, 128), (8192, 8192, 16), (8192, 8192, 32), (8192, 8192, 48), (8192, 8192, 4096 + 64), (8192, 8190, 0),
         (8188, 8192, 0), (8188, 8192, 16), (8176, 8192, 0), (8128, 8192, 0), (8000, 8192, 0), (8190, 8190, 0), (8190, 8190, 8), (8191, 8191, 0)]
for name in sys.argv[1].split(","):
    path = name if "/" in name else os.path.join(ROOT, "build", "explib", name)
    lib = ctypes.CDLL(path)
    lib.detexhipDecompressTextureLinearDevice.argtypes = [ctypes.c_uint32, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, ctypes.c_size_t, ctypes.c_uint32, vp, vp]
    for (W, H, pad) in cases:
        wb, hb = (W + 3) // 4, (H + 3) // 4
        pitch = W * px + pad
        data = ol.stream_u(fmt, wb * hb, seed=77)
        d = torch.from_numpy(np.ascontiguousarray(data)).cuda()
        out = torch.empty(H * pitch + 256, dtype=torch.uint8, device="cuda")
        step = lambda: lib.detexhipDecompressTextureLinearDevice(fmt.texture_format, d.data_ptr(), W, H, wb, hb, out.data_ptr(), pitch, F.native_pixel_format(fmt), st, None)
        if step() != 0:
            print(json.dumps({"lib": os.path.basename(path), "W": W, "H": H, "pitch": pitch, "error": True})); continue
        for _ in range(300): step()
        best = None
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(100): step()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 10
            best = us if best is None or us < best else best
        print(json.dumps({"lib": os.path.basename(path), "format": fmt.name, "W": W, "H": H, "pitch": pitch, "pitch_mod_64": pitch % 64, "us": round(best, 1)}), flush=True)
        del d, out
