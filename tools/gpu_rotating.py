#!/usr/bin/env python3
"""Launch time with the blocks coming out of HBM: the same decode over R different inputs in turn (R x blocks > 2.5 x the Infinity Cache),
beside the usual loop over ONE input (whose blocks are re-read from that cache) -- per library build, per format, per read-ahead mode.
usage: python tools/gpu_rotating.py LIB[,LIB...] FMT[,FMT...] [SIZE=8192] [MODES=1,2] [linear|tiled]        (bare library names: ab_libs/)
GPU box; one child process per library (DETEXHIP_LIB)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(names, size, modes, layout):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np, torch
    from detex_amd import binding, formats as F
    import streams
    binding.load()

    def steady(step, window=60):
        prev, done, us = None, 0, None
        for _ in range(12):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(window):
                step()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / window; done += window
            if prev is not None and done >= 300 and abs(us - prev) <= 0.012 * prev:
                break
            prev = us
        return us
    for name in names:
        fmt = F.BY_NAME[name]
        data = streams.make_stream("U", fmt, size // 4, size // 4)
        d = torch.from_numpy(np.ascontiguousarray(data)).cuda()
        out = torch.empty(size * size * fmt.pixel_bytes, dtype=torch.uint8, device="cuda")
        inputs = [d] + [torch.roll(d, 4096 * k) for k in range(1, max(3, -(-(640 << 20) // d.numel())))]
        alg = (size // 4) ** 2 * (fmt.block_bytes + 16 * fmt.pixel_bytes)
        row = {"lib": os.path.basename(binding.LIB_PATH), "format": name, "size": size, "layout": layout, "inputs": len(inputs)}

        def decode(src):
            if layout == "tiled":
                binding.decompress_tiled_device(fmt, src, size // 4, size // 4, out=out)
            else:
                binding.decompress_linear_device(fmt, src, size, size, out=out)
        row["one_input_us"] = round(steady(lambda: decode(d)), 2)
        k = [0]

        def rot():
            k[0] = (k[0] + 1) % len(inputs)
            decode(inputs[k[0]])
        for m in modes:
            binding.set_read_ahead(m)
            us = steady(rot)
            row["rotating_mode%d_us" % m] = round(us, 2); row["rotating_mode%d_frac" % m] = round(alg / (us * 1e-6) / 8e12, 4)
        binding.set_read_ahead(1)
        print(json.dumps(row), flush=True)
        del d, out, inputs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2].split(","), int(sys.argv[3]), [int(m) for m in sys.argv[4].split(",")], sys.argv[5])
    else:
        libs = sys.argv[1].split(",")
        size = sys.argv[3] if len(sys.argv) > 3 else "8192"
        modes = sys.argv[4] if len(sys.argv) > 4 else "1,2"
        layout = sys.argv[5] if len(sys.argv) > 5 else "linear"
        for lib in libs:
            path = lib if "/" in lib else os.path.join(ROOT, "ab_libs", lib)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", sys.argv[2], size, modes, layout], env=dict(os.environ, DETEXHIP_LIB=os.path.abspath(path)))
