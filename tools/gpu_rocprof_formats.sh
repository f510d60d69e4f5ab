#!/bin/bash
# rocprofv3 kernel-trace of the bench command for the headline formats: per-kernel duration of the LAST 200 of 1000 launches
# (past the power-management transient), to cross-check the steady-state figures of bench.py's per_format table.
#   bash tools/gpu_rocprof_formats.sh   ->  gpurun_out/rocprof_formats/<FORMAT>_last200.json + kernel_stats csv
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/rocprof_formats; rm -rf $OUT; mkdir -p $OUT
for fmt in ${@:-BC1 BC3 BPTC ETC2 ETC2_EAC BPTC_FLOAT BPTC_SIGNED_FLOAT}; do
  cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -T -d $OUT/p_$fmt -o k --output-format csv -- python $ROOT/bench.py --format $fmt --steps 200 --warmup 800 --no-cpu --no-extras > $OUT/$fmt.log 2>&1
  cd $ROOT
  python3 - "$fmt" "$OUT" <<'PY'
import csv, glob, json, sys
fmt, out = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/p_%s/**/*kernel_trace.csv" % fmt, recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if "decode_linear" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows][-200:]
d.sort()
res = {"format": fmt, "launches_traced": len(rows), "last200_mean_us": round(sum(d) / len(d) / 1e3, 2), "last200_median_us": round(d[len(d) // 2] / 1e3, 2),
       "last200_min_us": round(d[0] / 1e3, 2), "last200_max_us": round(d[-1] / 1e3, 2)}
json.dump(res, open(out + "/%s_last200.json" % fmt, "w")); print(res)
PY
  s=$(find $OUT/p_$fmt -name "*kernel_stats.csv" | head -1); [ -n "$s" ] && cp $s $OUT/${fmt}_kernel_stats.csv
  rm -rf $OUT/p_$fmt
done
