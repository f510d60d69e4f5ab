#!/usr/bin/env python3
"""pitch experiment: is the BC6H slowdown a power-of-two-stride (HBM channel) effect?"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from detex_amd import binding, formats as F
import oracle_lib as ol
W = H = 8192
n = (W // 4) * (H // 4)
for name in sys.argv[1:]:
    fmt = F.BY_NAME[name]; px = fmt.pixel_bytes
    d_blocks = torch.from_numpy(np.ascontiguousarray(ol.stream_u(fmt, n, seed=5))).cuda()
    for pad in (0, 128, 256, 512, 1024, 2048, 4096, 8192 + 256):
        pitch = W * px + pad
        d_out = torch.empty(H * pitch, dtype=torch.uint8, device="cuda")
        for _ in range(5): binding.decompress_linear_device(fmt, d_blocks, W, H, out=d_out, pitch=pitch)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(30): binding.decompress_linear_device(fmt, d_blocks, W, H, out=d_out, pitch=pitch)
        e1.record(); torch.cuda.synchronize()
        print("%-20s pitch = W*px + %-5d : %8.2f us" % (name, pad, e0.elapsed_time(e1) / 30 * 1e3), flush=True)
        del d_out
