#!/usr/bin/env python3
"""8f-3 measurement: a full mip chain decoded level by level (one launch each, what a caller of
detexLoadKTXFileWithMipmaps does today) vs detexhipDecompressLevelsLinearDevice (one launch)."""
import ctypes, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from detex_amd import binding, formats as F
import oracle_lib as ol

class Level(ctypes.Structure):
    _fields_ = [("d_blocks", ctypes.c_void_p), ("d_pixels", ctypes.c_void_p), ("pitch", ctypes.c_size_t),
                ("width", ctypes.c_int), ("height", ctypes.c_int), ("wb", ctypes.c_int), ("hb", ctypes.c_int)]

def main():
    lib = binding.load()
    lib.detexhipDecompressLevelsLinearDevice.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
    res = {}
    for name, top in (("BC1", 8192), ("BC1", 1024), ("BPTC", 2048)):
        fmt = F.BY_NAME[name]; px = fmt.pixel_bytes
        dims = []
        w = h = top
        while w >= 1: dims.append((w, h)); w >>= 1; h >>= 1
        d_in = [torch.from_numpy(ol.stream_u(fmt, ((w + 3) // 4) * ((h + 3) // 4), seed=i).copy()).cuda() for i, (w, h) in enumerate(dims)]
        d_out = [torch.empty(w * h * px, dtype=torch.uint8, device="cuda") for w, h in dims]
        arr = (Level * len(dims))(*[Level(d_in[i].data_ptr(), d_out[i].data_ptr(), w * px, w, h, (w + 3) // 4, (h + 3) // 4) for i, (w, h) in enumerate(dims)])
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        def per_level():
            for i, (w, h) in enumerate(dims):
                binding.decompress_linear_device(fmt, d_in[i], w, h, out=d_out[i])
        def one_launch():
            assert lib.detexhipDecompressLevelsLinearDevice(fmt.texture_format, arr, len(dims), F.native_pixel_format(fmt), stream, None) == 0
        out = {}
        for label, fn in (("per_level_launches", per_level), ("one_launch", one_launch)):
            for _ in range(5): fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(50): fn()
            e1.record(); torch.cuda.synchronize()
            out[label + "_us"] = round(e0.elapsed_time(e1) / 50 * 1e3, 2)
        out["levels"] = len(dims)
        res["%s/%d" % (name, top)] = out
        print(name, top, out, flush=True)
    print(json.dumps(res))

if __name__ == "__main__":
    main()
