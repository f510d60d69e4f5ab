#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/final; mkdir -p $OUT
time (timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err); python -c "
import json; d=json.load(open('$OUT/bench.json'))
print(d['value'], d['roofline']['frac'], d['host_tier'], d['per_format']['seconds'], d.get('strong_image_32768'))
for k,v in d['per_format']['formats'].items(): print(k, v['launch_us'], v['frac'], v['launches_before_reading'])"
timeout 900 python bench.py --no-cpu --no-extras --formats-json $OUT/formats_8192.json > /dev/null 2> $OUT/formats.err
timeout 900 python bench.py --no-cpu --no-extras --layout tiled --formats-json $OUT/formats_8192_tiled.json > /dev/null 2> $OUT/formats_tiled.err
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
