#!/bin/bash
# Round 4, BC7 attempt: parity, same-run A/B against the library at the previous commit (ab_libs/libdetexhip_prev.so), SQ counters per wave.
#   bash tools/gpu_round4_bc7.sh TAG [FORMATS=BPTC] [STREAMS=U,M,C]
set -u
export TMPDIR=/tmp
TAG=$1; FMTS=${2:-BPTC}; STREAMS=${3:-U,M,C}
ROOT=$(pwd); OUT=gpurun_out/r04_$TAG; rm -rf $OUT; mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py tests/test_quirks.py -m gpu -x -q -k "BPTC or bptc or quirk or fuzz or stream" 2>&1 | tail -3 > $OUT/parity.txt; cat $OUT/parity.txt
LIBS=${BASE_LIB:-ab_libs/libdetexhip_prev.so},detex_amd/lib/libdetexhip.so
python tools/gpu_ab.py --libs $LIBS --formats $FMTS --streams $STREAMS --rounds 3 --clocks --out $OUT/ab_linear.jsonl 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); c=d.get('clocks',{}); print('linear', d['lib'].split('/')[-1], d['format'], d['stream'], d['us'], d.get('rounds'), c.get('sclk_MHz_mean'), c.get('W_mean'))"
python tools/gpu_ab.py --libs $LIBS --formats $FMTS --streams $STREAMS --layout tiled --rounds 3 --out $OUT/ab_tiled.jsonl 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('tiled ', d['lib'].split('/')[-1], d['format'], d['stream'], d['us'], d.get('rounds'))"
for FMT in ${FMTS//,/ }; do for lib in ab_libs/libdetexhip_prev.so detex_amd/lib/libdetexhip.so; do
  n=$(basename $lib .so)
  cd /tmp && DETEXHIP_LIB=$ROOT/$lib timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace -d $ROOT/$OUT/sq_${FMT}_$n -o sq --output-format csv -- python $ROOT/tools/gpu_run_case.py $FMT U 8192 8192 0 6 linear > $ROOT/$OUT/sq_${FMT}_$n.log 2>&1
  cd $ROOT; f=$(find $OUT/sq_${FMT}_$n -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "$FMT $n" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "decode_linear" in r["Kernel_Name"]]
d = collections.defaultdict(list)
for r in rows: d[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sorted(v)[len(v)//2] for k, v in d.items()}
w = m.get("SQ_WAVES", 1)
print(sys.argv[2], "per wave:", {k: round(v / w, 1) for k, v in m.items() if k != "SQ_WAVES"}, "waves", int(w))
PY
done; done | tee $OUT/sq_per_wave.txt
