#!/usr/bin/env python3
"""GPU box: the reference fill (tools/ubench/hbm_ref.hip) with each of the eight store cache policies (sc0 / sc1 / nt), 256 MiB and 1 GiB.
usage: python tools/gpu_store_policy.py [out.jsonl]"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, hbmref
lib = hbmref.load()
lib.hbmref_fill_policy.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p]
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
out = open(sys.argv[1], "a") if len(sys.argv) > 1 else None
names = ["(none)", "sc0", "sc1", "sc0 sc1", "nt", "sc0 nt", "sc1 nt", "sc0 sc1 nt"]
for mib in (256, 1024):
    nbytes = mib << 20
    buf = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    for rnd in range(2):
        for policy in range(8):
            seed = [1]
            def step():
                seed[0] += 1
                if lib.hbmref_fill_policy(buf.data_ptr(), nbytes, policy, seed[0], st) != 0:
                    raise RuntimeError("fill failed")
            us = hbmref.time_us(step, launches=40, warmup=8)
            row = {"op": "fill_policy", "MiB": mib, "policy": names[policy], "round": rnd, "us": round(us, 2), "GBps": round(nbytes / us / 1e3, 1)}
            print(json.dumps(row), flush=True)
            if out:
                out.write(json.dumps(row) + "\n")
    del buf
