#!/usr/bin/env python3
"""Large textures through the HOST-pointer drop-in entry (detexDecompressTextureLinear on pageable memory): milliseconds per call, per library
build -- the duplex staged path (uploads beside downloads, host_tier.cpp: via_staging_duplex) against a build without it.
usage: python tools/gpu_host_big.py LIB[,LIB...] [FMT:SIDE ...]      (bare library names: ab_libs/)     GPU box; one child process per library."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = ["BC1:8192", "BC1:4096", "BPTC:8192", "BPTC_FLOAT:4096", "RGTC1:16384", "RGTC2:8192", "BC1:16384"]


def child(cases):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as ol
    from detex_amd import binding, formats as F
    import streams
    binding.load()
    api = ol.DetexAPI(binding.LIB_PATH)
    for case in cases:
        name, side = case.split(":"); side = int(side)
        fmt = F.BY_NAME[name]
        data = np.ascontiguousarray(streams.make_stream("U", fmt, side // 4, side // 4))
        good = data
        if name == "BPTC":                                          # (valid blocks only: the call's result is then true)
            good = data.copy().reshape(-1, 16); good[:, 0] |= 1; good = good.reshape(-1)
        out = np.zeros(side * side * fmt.pixel_bytes, np.uint8)
        for _ in range(2):
            ok, _ = api.linear(fmt, good, side, side, out=out)
        t = []
        for _ in range(9):
            t0 = time.perf_counter()
            ok, _ = api.linear(fmt, good, side, side, out=out)
            t.append((time.perf_counter() - t0) * 1e3)
        t.sort()
        mib_in, mib_out = good.size / 2 ** 20, out.size / 2 ** 20
        print(json.dumps({"lib": os.path.basename(binding.LIB_PATH), "format": name, "side": side, "ok": bool(ok), "blocks_MiB": round(mib_in, 1), "pixels_MiB": round(mib_out, 1),
                          "median_ms": round(t[len(t) // 2], 3), "best_ms": round(t[0], 3), "Gpixel_per_s": round(side * side / t[len(t) // 2] / 1e6, 2),
                          "link_GBps_both_ways": round((good.size + out.size) / t[len(t) // 2] / 1e6, 1)}), flush=True)
        del out, data, good


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2:])
    else:
        cases = sys.argv[2:] or DEFAULT
        for lib in sys.argv[1].split(","):
            path = lib if "/" in lib else os.path.join(ROOT, "ab_libs", lib)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child"] + cases, env=dict(os.environ, DETEXHIP_LIB=os.path.abspath(path)))
