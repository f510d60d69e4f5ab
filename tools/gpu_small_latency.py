#!/usr/bin/env python3
"""Latency of small calls through the host-pointer tier (the reference's own entry points): detexDecompressTextureLinear on
64^2 .. 2048^2 BC1 / BC7 textures and the one-block leaf function, per library build, beside the compiled reference on one
host thread.  usage: python tools/gpu_small_latency.py lib1[,lib2...]   (bare names: build/explib/)"""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch  # noqa: F401  (torch first: its HIP runtime)
from detex_amd import formats as F
import oracle_lib as ol

def per_call_us(fn, budget_s=0.25, min_calls=20):
    fn(); fn()
    t0 = time.perf_counter(); n = 0
    while n < min_calls or time.perf_counter() - t0 < budget_s:
        fn(); n += 1
    return (time.perf_counter() - t0) / n * 1e6

libs = []
for name in sys.argv[1].split(","):
    path = name if "/" in name else os.path.join(ROOT, "build", "explib", name)
    libs.append((os.path.basename(path), ol.DetexAPI(path)))
if ol.have_ref():
    libs.append(("reference (1 host thread)", ol.load_ref()))
for fname in ("BC1", "BPTC"):
    fmt = F.BY_NAME[fname]
    blk = ol.stream_u(fmt, 1, seed=5)
    for label, api in libs:
        row = {"lib": label, "format": fname}
        fn = api.block_fn(fmt)
        out = np.zeros(16 * fmt.pixel_bytes, np.uint8)
        row["one_block_us"] = round(per_call_us(lambda: fn(ol._ptr(blk), 0xFFFFFFFF, 0, ol._ptr(out))), 2)
        for side in (64, 128, 256, 512, 1024, 2048):
            data = ol.stream_u(fmt, (side // 4) ** 2, seed=side)
            o = np.empty(side * side * fmt.pixel_bytes, np.uint8)
            row["%dx%d_us" % (side, side)] = round(per_call_us(lambda: api.linear(fmt, data, side, side, out=o)), 1)
        print(json.dumps(row), flush=True)
