#!/usr/bin/env python3
"""Same-process A/B of library builds: steady-state launch time of the linear (or block-major) decode kernel.

usage: python tools/gpu_ab.py --libs detex_amd/lib/libdetexhip.so,build/explib/libdetexhip_exp_prio1.so[:variant] --formats BPTC --streams U,M,C
                              [--layout linear|tiled] [--size 8192] [--height H] [--rounds 2] [--out gpurun_out/ab.jsonl] [--clocks]
Every library is dlopen'ed once (RTLD_LOCAL: the builds share symbol names) and the configurations are visited round-robin
(A B C A B C ...), so clock / thermal drift over the run hits all of them alike.  One JSON line per (library, format, stream)
with the per-round times; `us` is the minimum over rounds of the settled window average.  --clocks samples rocm-smi while a
configuration runs.  No verification here (pytest -m gpu / bench.py do that)."""
import argparse, ctypes, json, os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from detex_amd import formats as F
import streams

ap = argparse.ArgumentParser()
ap.add_argument("--libs", required=True)
ap.add_argument("--formats", default="BPTC")
ap.add_argument("--streams", default="U")
ap.add_argument("--layout", default="linear")
ap.add_argument("--size", type=int, default=8192)
ap.add_argument("--height", type=int, default=0)
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--out", default=None)
ap.add_argument("--clocks", action="store_true")
args = ap.parse_args()
W = args.size; H = args.height or args.size

vp = ctypes.c_void_p
libs = []
for spec in args.libs.split(","):
    name, _, variant = spec.partition(":")
    path = name if "/" in name else os.path.join(ROOT, "build", "explib", name)   # bare names: measurement builds
    lib = ctypes.CDLL(path)
    lib.detexhipDecompressTextureLinearDevice.argtypes = [ctypes.c_uint32, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, ctypes.c_size_t, ctypes.c_uint32, vp, vp]
    lib.detexhipDecompressTextureTiledDevice.argtypes = [ctypes.c_uint32, vp, ctypes.c_int, ctypes.c_int, vp, ctypes.c_uint32, vp, vp]
    lib.detexhipSetKernelVariant.argtypes = [ctypes.c_int]
    lib.detexGetErrorMessage.restype = ctypes.c_char_p
    libs.append((spec, lib, int(variant or 0)))

def poll_clocks(stop, samples):
    while not stop[0]:
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
            j = json.loads(o); c = j[sorted(j)[0]]
            s = {}
            for k, v in c.items():
                if "sclk clock speed" in k: s["sclk"] = v.strip("()")
                if "ower" in k and "(W)" in k: s["W"] = v
            samples.append(s)
        except Exception as e:  # noqa
            samples.append({"err": str(e)[:60]})
        time.sleep(0.05)

out = open(args.out, "a") if args.out else None
st = vp(torch.cuda.current_stream().cuda_stream)
for fname in args.formats.split(","):
    fmt = F.BY_NAME[fname]
    pf = F.native_pixel_format(fmt)
    for kind in args.streams.split(","):
        data = streams.make_stream(kind, fmt, W // 4, H // 4)
        if data is None:
            continue
        d_blocks = torch.from_numpy(np.ascontiguousarray(data)).cuda()
        d_out = torch.empty(W * H * fmt.pixel_bytes, dtype=torch.uint8, device="cuda")
        alg = (W // 4) * (H // 4) * (fmt.block_bytes + 16 * fmt.pixel_bytes)
        results = {spec: [] for spec, _, _ in libs}
        clocks = {spec: [] for spec, _, _ in libs}
        for rnd in range(args.rounds):
            for spec, lib, variant in libs:
                lib.detexhipSetKernelVariant(variant)
                if args.layout == "tiled":
                    step = lambda: lib.detexhipDecompressTextureTiledDevice(fmt.texture_format, d_blocks.data_ptr(), W // 4, H // 4, d_out.data_ptr(), pf, st, None)
                else:
                    step = lambda: lib.detexhipDecompressTextureLinearDevice(fmt.texture_format, d_blocks.data_ptr(), W, H, W // 4, H // 4, d_out.data_ptr(), W * fmt.pixel_bytes, pf, st, None)
                if step() != 0:
                    print("launch failed:", spec, lib.detexGetErrorMessage(), file=sys.stderr); continue
                stop, samples = [False], []
                if args.clocks:
                    t = threading.Thread(target=poll_clocks, args=(stop, samples)); t.start()
                window = 100 if alg < 1.5e9 else 25
                prev, done, us, spent_ms = None, 0, None, 0.0
                for _ in range(40):
                    n = window if us is None else max(window, int(4000.0 / us) + 1)     # windows of >= 4 ms
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(n):
                        step()
                    e1.record(); torch.cuda.synchronize()
                    ms = e0.elapsed_time(e1); us = ms / n * 1e3; done += n; spent_ms += ms
                    if prev is not None and done >= 6 * window and spent_ms >= 40.0 and abs(us - prev) <= 0.012 * prev:
                        break
                    if spent_ms > 400.0:
                        break
                    prev = us
                if args.clocks:
                    # keep the kernel running for ~3 s so the sampler (one rocm-smi call takes ~0.3 s) sees the settled state
                    t_end = time.time() + 3.0
                    while time.time() < t_end:
                        for _ in range(200):
                            step()
                        torch.cuda.synchronize()
                    stop[0] = True; t.join()
                    tail = [x for x in samples[len(samples) // 2:] if "sclk" in x]
                    mhz = [int("".join(ch for ch in x["sclk"] if ch.isdigit())) for x in tail]
                    watts = [float(x["W"]) for x in tail if "W" in x]
                    clocks[spec] = {"samples": len(tail), "sclk_MHz_mean": round(sum(mhz) / max(1, len(mhz))), "sclk_MHz_min": min(mhz or [0]), "sclk_MHz_max": max(mhz or [0]),
                                    "W_mean": round(sum(watts) / max(1, len(watts))), "W_max": max(watts or [0])}
                results[spec].append(round(us, 2))
        for spec, _, _ in libs:
            if not results[spec]:
                continue
            best = min(results[spec])
            row = {"lib": spec, "format": fname, "stream": kind, "layout": args.layout, "size": [W, H], "us": best,
                   "frac": round(alg / (best * 1e-6) / 8e12, 4), "rounds": results[spec]}
            if args.clocks:
                row["clocks"] = clocks[spec]
            line = json.dumps(row)
            print(line, flush=True)
            if out:
                out.write(line + "\n"); out.flush()
        del d_blocks, d_out
        torch.cuda.empty_cache()
