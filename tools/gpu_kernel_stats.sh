#!/bin/bash
# rocprofv3 --kernel-trace --stats summaries for profiles/: the headline workload alone (so that the decode kernel's average is the
# 8192^2 BC1 launch and nothing else) and the driver's exact command (every kernel of the extras under its full name).
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/kernel_stats; rm -rf $OUT; mkdir -p $OUT
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/headline -o k --output-format csv -- python $ROOT/bench.py --steps 100 --warmup 10 --no-cpu --no-extras --no-host-tier > $OUT/headline_bench.json 2> $OUT/headline.log
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/driver -o k --output-format csv -- python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_bench.json 2> $OUT/driver.log
cd $ROOT
for t in headline driver; do f=$(find $OUT/$t -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/${t}_kernel_stats.csv; rm -rf $OUT/$t; done
head -3 $OUT/headline_kernel_stats.csv | cut -c1-250
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/kernel_stats/headline_bench.json").read().strip().splitlines()[-1])
print("bench line of the same run: launch_us", d["roofline"]["launch_us"], "ms_per_step", d["ms_per_step"], "settle", d["settle"]["launches_before_warmup"])
PY
