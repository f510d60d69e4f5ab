#!/usr/bin/env python3
"""Fill of 256 MiB / 1 GiB with the product's store policy in four per-instruction shapes (tools/ubench/hbm_ref.hip: fill_shape_kernel):
what would a block-major kernel lose if it transposed inside quads of lanes (64-byte pieces) instead of through LDS (1 KiB runs)?
usage (GPU box): python tools/gpu_store_shape.py"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import hbmref
lib = hbmref.load()
lib.hbmref_fill_shape.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p]
names = {0: "1 KiB runs", 1: "64 B pieces, 256 B apart", 2: "16 B pieces, 64 B apart", 3: "128 B lines, 512 B apart"}
st = torch.cuda.current_stream().cuda_stream
for mib in (256, 1024):
    nbytes = mib << 20
    buf = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    for rnd in range(3):
        for shape in (0, 1, 3, 2):
            seed = [1]
            def step():
                seed[0] += 1
                assert lib.hbmref_fill_shape(buf.data_ptr(), nbytes, shape, seed[0], st) == 0
            us = hbmref.time_us(step, launches=200, warmup=300 if rnd == 0 and shape == 0 else 20)
            print(json.dumps({"op": "fill_shape", "MiB": mib, "shape": names[shape], "round": rnd, "us": round(us, 2), "GBps": round(nbytes / us / 1e3, 1)}), flush=True)
