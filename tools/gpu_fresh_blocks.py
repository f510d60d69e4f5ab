#!/usr/bin/env python3
"""Where do a decode's blocks come from when they were WRITTEN just before it?  Launch time of the linear decode (kernel alone, HIP events)
with the blocks produced, right before each launch, into one of R rotating device buffers (R x blocks >= 640 MiB) by
  upload   a copy out of pinned host memory on the same stream (the upload -> decode pipeline of a texture streamer)
  d2d      a device-to-device copy on the same stream (blocks produced by another kernel)
  upload_elsewhere / d2d_elsewhere   the same copies into a buffer the launch does NOT read (control: the gap and the copy's own traffic, blocks cold)
beside
  cold     the same rotating buffers with nothing written in between (every block out of HBM: bench.py's blocks_from_hbm)
  one      ONE input again and again (its blocks re-read from the Infinity Cache: the contract's loop)
usage: python tools/gpu_fresh_blocks.py FMT[,FMT...] [SIZE=8192] [STEPS=80]     GPU box."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(names, size, steps):
    import numpy as np, torch
    from detex_amd import binding, formats as F
    import streams
    binding.load()
    for name in names:
        fmt = F.BY_NAME[name]
        data = np.ascontiguousarray(streams.make_stream("U", fmt, size // 4, size // 4))
        host = torch.from_numpy(data).pin_memory()
        d = host.cuda()
        out = torch.empty(size * size * fmt.pixel_bytes, dtype=torch.uint8, device="cuda")
        n_in = max(3, -(-(640 << 20) // d.numel()))
        inputs = [d.clone() for _ in range(n_in)]
        spare = d.clone()
        scratch = d.clone()
        alg = (size // 4) ** 2 * (fmt.block_bytes + 16 * fmt.pixel_bytes)

        def run(producer):
            def once(k, timed):
                src = inputs[k % n_in]
                if producer == "upload":
                    src.copy_(host, non_blocking=True)
                elif producer == "d2d":
                    src.copy_(spare, non_blocking=True)
                elif producer == "upload_elsewhere":
                    scratch.copy_(host, non_blocking=True)
                elif producer == "d2d_elsewhere":
                    scratch.copy_(spare, non_blocking=True)
                elif producer == "one":
                    src = d
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                binding.decompress_linear_device(fmt, src, size, size, out=out)
                e1.record()
                if timed is not None:
                    timed.append((e0, e1))
            for k in range(30):                 # settle (clocks, power management)
                once(k, None)
            torch.cuda.synchronize()
            ev = []
            for k in range(steps):
                once(k, ev)
            torch.cuda.synchronize()
            us = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
            return us[len(us) // 2], us[len(us) // 10], us[-len(us) // 10]
        row = {"format": name, "size": size, "inputs": n_in, "block_MiB": round(d.numel() / 2 ** 20, 1), "steps": steps}
        for mode in (1,):
            binding.set_read_ahead(mode)
            for producer in ("one", "cold", "upload", "upload_elsewhere", "d2d", "d2d_elsewhere", "one", "cold"):
                med, lo, hi = run(producer)
                key = producer if producer not in row else producer + "_again"
                row[key] = {"median_us": round(med, 2), "p10_us": round(lo, 2), "p90_us": round(hi, 2), "frac": round(alg / (med * 1e-6) / 8e12, 4)}
        print(json.dumps(row), flush=True)
        del d, out, inputs, spare, host
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main(sys.argv[1].split(","), int(sys.argv[2]) if len(sys.argv) > 2 else 8192, int(sys.argv[3]) if len(sys.argv) > 3 else 80)
