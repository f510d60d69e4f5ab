#!/usr/bin/env python3
"""HBM traffic per launch from rocprofv3 PMC counters, for a list of (format, layout[, side]) kernels (side 8192 unless given).

Runs on the GPU box (inside gpurun).  Per the guide (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE in
SEPARATE --pmc passes together with --kernel-trace only; on gfx950 FETCH_SIZE (KiB) counts the 128-byte requests of a
wide coalesced streaming read at 64 bytes, so it is doubled; cross-checked with TCC_EA0_RDREQ_sum x 128 B and
TCC_EA0_WRREQ_sum x 64 B in a third pass.  Writes OUT/pmc_traffic.json (its entries are merged into profiles/pmc_traffic.json, which
bench.py replays where it cannot measure) and keeps the raw counter rows of the decode kernels as CSV next to it.
usage: python tools/pmc_traffic.py OUTDIR FMT:layout[:side] [FMT:layout[:side] ...]"""
import csv, glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from detex_amd import formats as F

out_dir = os.path.abspath(sys.argv[1]); os.makedirs(out_dir, exist_ok=True)
jobs = [(a.split(":") + ["8192"])[:3] for a in sys.argv[2:]]
PASSES = {"FETCH_SIZE": ["FETCH_SIZE"], "WRITE_SIZE": ["WRITE_SIZE"], "EA": ["TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum"]}
result = {}
for name, layout, side in jobs:
    side = int(side)
    fmt = F.BY_NAME[name]
    kernel = "decode_linear" if layout == "linear" else "decode_blocks"
    vals = {}
    for tag, ctrs in PASSES.items():
        d = os.path.join(out_dir, "pmc_%s_%s_%d_%s" % (name, layout, side, tag))
        cmd = ["rocprofv3", "--pmc"] + ctrs + ["--kernel-trace", "-T", "-d", d, "-o", "p", "--output-format", "csv", "--",
               sys.executable, os.path.join(ROOT, "bench.py"), "--format", name, "--layout", layout, "--size", str(side), "--steps", "6", "--warmup", "2", "--no-cpu", "--no-extras"]
        subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        rows = [r for f in files for r in csv.DictReader(open(f)) if kernel in r["Kernel_Name"]]
        keep = os.path.join(out_dir, "%s_%s%s_pmc_%s.csv" % (name.lower(), layout, "" if side == 8192 else "_%d" % side, tag))
        if rows:
            with open(keep, "w", newline="") as fh:
                w = csv.DictWriter(fh, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows[:24])
        for c in ctrs:
            v = sorted(float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == c)
            vals[c] = v[len(v) // 2] if v else None
    blocks = (side // 4) ** 2
    alg = blocks * (fmt.block_bytes + 16 * fmt.pixel_bytes)
    if vals.get("FETCH_SIZE") is None or vals.get("WRITE_SIZE") is None:
        print(name, layout, "counters missing", vals); continue
    fetch = vals["FETCH_SIZE"] * 1024 * 2          # gfx950: 128-byte requests tallied at 64 B
    write = vals["WRITE_SIZE"] * 1024
    ent = {"hbm_bytes_per_launch": int(fetch + write), "fetch_bytes": int(fetch), "write_bytes": int(write),
           "algorithmic_bytes_per_launch": alg, "ratio": round((fetch + write) / alg, 4),
           "ea_rdreq_x128": None if vals.get("TCC_EA0_RDREQ_sum") is None else int(vals["TCC_EA0_RDREQ_sum"] * 128),
           "ea_wrreq_x64": None if vals.get("TCC_EA0_WRREQ_sum") is None else int(vals["TCC_EA0_WRREQ_sum"] * 64),
           "source": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum (three separate passes with --kernel-trace only; median of the "
                     "profiled %s launches); FETCH_SIZE KiB x 2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE KiB" % kernel}
    result["%s/%d/%s" % (name, side, layout)] = ent
    print(name, layout, ent["ratio"], ent["hbm_bytes_per_launch"], alg, flush=True)
json.dump(result, open(os.path.join(out_dir, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
