#!/usr/bin/env python3
"""ad-hoc GPU timing experiments: python tools/gpu_exp.py FORMAT [FORMAT...]; each format is timed on
several input streams to separate kernel cost from data effects."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from detex_amd import binding, formats as F
import oracle_lib as ol, streams
W = H = 8192
n = (W // 4) * (H // 4)
def timeit(fmt, d_blocks, d_out, steps=30, variant=0):
    binding.set_kernel_variant(variant)
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    for _ in range(5): binding.decompress_linear_device(fmt, d_blocks, W, H, out=d_out, status=status)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(steps): binding.decompress_linear_device(fmt, d_blocks, W, H, out=d_out, status=status)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3
for name in sys.argv[1:]:
    fmt = F.BY_NAME[name]
    d_out = torch.empty(W * H * fmt.pixel_bytes, dtype=torch.uint8, device="cuda")
    datasets = {"U(seed A)": ol.stream_u(fmt, n, seed=0xD37E5005), "U(seed B)": ol.stream_u(fmt, n, seed=0xD37E501A),
                "M": streams.stream_m(fmt, ol.stream_u(fmt, n, seed=3)), "zeros": np.zeros(n * fmt.block_bytes, np.uint8)}
    if fmt.name.startswith("BPTC_"):
        for m in (0, 1, 5, 10, 13):
            b = ol.stream_u(fmt, n, seed=100 + m).reshape(-1, 16).copy()
            code = streams.BC6H_MODE_CODES[m]
            b[:, 0] = (b[:, 0] & np.uint8(0xFC if m < 2 else 0xE0)) | np.uint8(code)
            datasets["mode%d only" % m] = b.reshape(-1)
    for label, data in datasets.items():
        d_b = torch.from_numpy(np.ascontiguousarray(data)).cuda()
        t = timeit(fmt, d_b, d_out)
        t3 = timeit(fmt, d_b, d_out, variant=3) if fmt.name.startswith("BPTC_") else float("nan")
        print("%-20s %-14s %8.2f us   (variant 3: %8.2f us)" % (name, label, t, t3), flush=True)
