#!/usr/bin/env python3
"""The whole 32768^2 image through ONE call of the device entry (which bands it behind the read-ahead pass), per library build: steady launch time.
usage: python tools/gpu_whole_image.py LIB[,LIB...] [FMT[,FMT]] [rounds]      (bare library names: ab_libs/)     GPU box; one child per library."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(names, rounds):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np, torch
    from detex_amd import binding, formats as F
    import streams
    binding.load()
    side = 32768
    for name in names:
        fmt = F.BY_NAME[name]
        d = torch.from_numpy(np.ascontiguousarray(streams.make_stream("U", fmt, side // 4, side // 4))).cuda()
        out = torch.empty(side * side * fmt.pixel_bytes, dtype=torch.uint8, device="cuda")
        alg = (side // 4) ** 2 * (fmt.block_bytes + 16 * fmt.pixel_bytes)
        for _ in range(15):
            binding.decompress_linear_device(fmt, d, side, side, out=out)
        torch.cuda.synchronize()
        us = []
        for _ in range(rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                binding.decompress_linear_device(fmt, d, side, side, out=out)
            e1.record(); torch.cuda.synchronize()
            us.append(e0.elapsed_time(e1) * 100.0)
        us.sort()
        print(json.dumps({"lib": os.path.basename(binding.LIB_PATH), "format": name, "side": side, "median_us": round(us[len(us) // 2], 1), "min_us": round(us[0], 1),
                          "frac": round(alg / (us[len(us) // 2] * 1e-6) / 8e12, 4)}), flush=True)
        del d, out
        torch.cuda.empty_cache()


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2].split(","), int(sys.argv[3]))
    else:
        fmts = sys.argv[2] if len(sys.argv) > 2 else "BC1,BPTC_FLOAT"
        rounds = sys.argv[3] if len(sys.argv) > 3 else "5"
        for lib in sys.argv[1].split(","):
            path = lib if "/" in lib else os.path.join(ROOT, "ab_libs", lib)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", fmts, rounds], env=dict(os.environ, DETEXHIP_LIB=os.path.abspath(path)))
