#!/usr/bin/env python3
"""Launch one decode configuration a few times (the command rocprofv3 wraps for counter passes).
usage: python tools/gpu_run_case.py FORMAT STREAM WIDTH HEIGHT [PITCH_PAD=0] [LAUNCHES=8] [LAYOUT=linear] [fill]
`fill` as last argument launches the image-layout reference fill over the same output image instead of the decoder."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
from detex_amd import binding, formats as F
import oracle_lib as ol, streams

fmt = F.BY_NAME[sys.argv[1]]; kind = sys.argv[2]; W = int(sys.argv[3]); H = int(sys.argv[4])
pad = int(sys.argv[5]) if len(sys.argv) > 5 else 0
n = int(sys.argv[6]) if len(sys.argv) > 6 else 8
layout = sys.argv[7] if len(sys.argv) > 7 else "linear"
fill = sys.argv[-1] == "fill"
px = fmt.pixel_bytes
wb, hb = (W + 3) // 4, (H + 3) // 4
pitch = W * px + pad
out = torch.empty(H * pitch + 256, dtype=torch.uint8, device="cuda")
if fill:
    import hbmref
    lib = hbmref.load()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(n + 3):
        assert lib.hbmref_fill_image(out.data_ptr(), W * px, H, pitch, 2, 7, st) == 0
else:
    data = streams.make_stream(kind, fmt, wb, hb) if (W % 4 == 0 and H % 4 == 0) else ol.stream_u(fmt, wb * hb)
    d = torch.from_numpy(np.ascontiguousarray(data)).cuda()
    binding.load()
    for _ in range(n + 3):
        if layout == "tiled":
            binding.decompress_tiled_device(fmt, d, wb, hb, out=out)
        else:
            binding.decompress_linear_device(fmt, d, W, H, out=out, pitch=pitch)
torch.cuda.synchronize()
