#!/bin/bash
# same-run comparison of the round-1 library (built from commit 0ebf32e into build/explib/libdetexhip_r01.so) with the current one
export TMPDIR=/tmp
OUT=gpurun_out/cmp_r01; mkdir -p $OUT; ROOT=$(pwd)
FM=BPTC,ETC2,ETC2_EAC,ETC2_PUNCHTHROUGH,SIGNED_RGTC2,SIGNED_RGTC1,EAC_SIGNED_R11,EAC_R11,RGTC2,RGTC1,BPTC_FLOAT,BPTC_SIGNED_FLOAT,BC1,BC3
rm -f $OUT/times.jsonl
for rep in 1 2; do for lib in libdetexhip_r01 libdetexhip; do
DETEXHIP_LIB=$ROOT/detex_amd/lib/$lib.so timeout 400 python tools/gpu_time.py $FM U linear 8192 $lib 2>>$OUT/err.log >> $OUT/times.jsonl
done; done
python3 - <<PY
import json,collections
d=collections.defaultdict(lambda: collections.defaultdict(list))
for l in open("$OUT/times.jsonl"):
    j=json.loads(l); d[j['format']][j['lib']].append(j['launch_us'])
for f,v in d.items(): print("%-20s r01 %s  now %s" % (f, v['libdetexhip_r01'], v['libdetexhip']))
PY
