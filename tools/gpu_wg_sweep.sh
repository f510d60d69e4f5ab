#!/bin/bash
# Resident workgroups per CU of the linear kernels (dynamic-LDS occupancy cap, launchers.h: occupancy_cap_lds): every format, streams U / C
# (M for the BPTC formats), caps 3..7 and none (wg0).  bash tools/build_exp_libs.sh wg0 wg3 wg4 wg5 wg6 wg7 first.
set -u
OUT=${1:-gpurun_out/wg_sweep}; mkdir -p $OUT
E=build/explib
L=$E/libdetexhip_exp_wg0.so,$E/libdetexhip_exp_wg3.so,$E/libdetexhip_exp_wg4.so,$E/libdetexhip_exp_wg5.so,$E/libdetexhip_exp_wg6.so,$E/libdetexhip_exp_wg7.so
rm -f $OUT/wg_8192.jsonl $OUT/wg_other.jsonl
python tools/gpu_ab.py --libs $L --formats BC1,BC1A,BC2,BC3,RGTC1,SIGNED_RGTC1,RGTC2,SIGNED_RGTC2,ETC1,ETC2,ETC2_PUNCHTHROUGH,ETC2_EAC,EAC_R11,EAC_SIGNED_R11,EAC_RG11,EAC_SIGNED_RG11 --streams U,C --rounds 2 --out $OUT/wg_8192.jsonl 2>/dev/null > /dev/null
python tools/gpu_ab.py --libs $L --formats BPTC,BPTC_FLOAT,BPTC_SIGNED_FLOAT --streams U,M,C --rounds 2 --out $OUT/wg_8192.jsonl 2>/dev/null > /dev/null
python tools/gpu_ab.py --libs $L --formats BC1,BPTC,BPTC_FLOAT --streams U,C --size 16384 --height 8192 --rounds 2 --out $OUT/wg_other.jsonl 2>/dev/null > /dev/null
python tools/gpu_ab.py --libs $L --formats BC1,BC3,BPTC,BPTC_FLOAT --streams U,C --size 2048 --rounds 2 --out $OUT/wg_other.jsonl 2>/dev/null > /dev/null
python3 - $OUT <<'PY'
import json, sys, collections
for name in ("wg_8192.jsonl", "wg_other.jsonl"):
    t = collections.OrderedDict()
    for l in open(sys.argv[1] + "/" + name):
        d = json.loads(l)
        key = "%-18s %s %5dx%-5d" % (d["format"], d["stream"], d["size"][0], d["size"][1])
        t.setdefault(key, {})[d["lib"].split("_wg")[-1].split(".")[0]] = d["us"]
    print(name); print("%-34s" % "format stream size", "  ".join("wg%s" % k for k in ("0", "7", "6", "5", "4", "3")), " best")
    for key, v in t.items():
        best = min(v, key=v.get)
        print("%-34s" % key, "  ".join("%6.2f" % v.get(k, 0) for k in ("0", "7", "6", "5", "4", "3")), "  wg%s %+.1f%%" % (best, 100 * (v[best] / v["0"] - 1)))
PY
