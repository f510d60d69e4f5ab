#!/usr/bin/env python3
"""Steady-state launch time of the linear (or tiled) decode kernel for a list of formats / streams at 8192^2.
usage: [DETEXHIP_LIB=...] python tools/gpu_time.py FMT[,FMT...] [STREAMS=U] [layout=linear] [size=8192] [tag]
One JSON line per (format, stream); no verification (use bench.py / pytest for that)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from detex_amd import binding, formats as F
import streams

names = sys.argv[1].split(",")
kinds = (sys.argv[2] if len(sys.argv) > 2 else "U").split(",")
layout = sys.argv[3] if len(sys.argv) > 3 else "linear"
size = int(sys.argv[4]) if len(sys.argv) > 4 else 8192
tag = sys.argv[5] if len(sys.argv) > 5 else os.path.basename(binding.LIB_PATH)
binding.load()
for name in names:
    fmt = F.BY_NAME[name]
    for kind in kinds:
        data = streams.make_stream(kind, fmt, size // 4, size // 4)
        if data is None:
            continue
        d_blocks = torch.from_numpy(np.ascontiguousarray(data)).cuda()
        d_out = torch.empty(size * size * fmt.pixel_bytes, dtype=torch.uint8, device="cuda")
        step = (lambda: binding.decompress_tiled_device(fmt, d_blocks, size // 4, size // 4, out=d_out)) if layout == "tiled" else \
               (lambda: binding.decompress_linear_device(fmt, d_blocks, size, size, out=d_out))
        prev, done, hist = None, 0, []
        for _ in range(14):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100):
                step()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 10.0; done += 100; hist.append(round(us, 1))
            if prev is not None and done >= 600 and abs(us - prev) <= 0.012 * prev:
                break
            prev = us
        alg = (size // 4) ** 2 * (fmt.block_bytes + 16 * fmt.pixel_bytes)
        print(json.dumps({"lib": tag, "format": name, "stream": kind, "layout": layout, "launch_us": round(us, 2),
                          "frac": round(alg / (us * 1e-6) / 8e12, 4), "windows": hist}), flush=True)
        del d_blocks, d_out
        torch.cuda.empty_cache()
