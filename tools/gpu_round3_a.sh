#!/bin/bash
# Round 3, GPU call A: parity of the refactored library, HBM reference sweep (data patterns x sizes), BC7 / BC6H same-process A/B
# of the measurement builds (build/explib), clocks of decode-only / store-only builds.  Results: gpurun_out/r03a/.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r03a
rm -rf $OUT; mkdir -p $OUT
E=build/explib
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4 | tee $OUT/pytest_gpu.log
echo "== counters"; (rocprofv3 -L 2>/dev/null || rocprofv3 --list-avail 2>/dev/null) > $OUT/counters.txt 2>&1; grep -c . $OUT/counters.txt
grep -o -E "(TCC|TCP|GRBM|SQ|MALL|CPC|TA|TD)_[A-Za-z0-9_]+" $OUT/counters.txt | sort -u > $OUT/counter_names.txt; wc -l $OUT/counter_names.txt
echo "== HBM reference sweep"; timeout 600 python tools/gpu_hbm_ref.py $OUT/hbm_reference.jsonl 2>$OUT/hbm.err | cut -c1-170
echo "== BC7 A/B (8192^2, U/M/C)"
LIBS=$PWD/detex_amd/lib/libdetexhip.so,$E/libdetexhip_r02.so,$E/libdetexhip_exp_plain.so,$E/libdetexhip_exp_waves8.so,$E/libdetexhip_exp_waves8+grp1.so,$E/libdetexhip_exp_waves8+grp2.so,$E/libdetexhip_exp_waves7+grp2.so,$E/libdetexhip_exp_prio1.so,$E/libdetexhip_exp_prio2.so,$E/libdetexhip_exp_prio3.so,$E/libdetexhip_exp_plain+prio1.so
timeout 900 python tools/gpu_ab.py --libs $LIBS --formats BPTC --streams U,M,C --rounds 2 --out $OUT/bc7_ab.jsonl 2>>$OUT/ab.err | cut -c1-200
echo "== BC7 persistent grids (A/B build variants 6 / 7), with and without priority staging"
timeout 600 python tools/gpu_ab.py --libs $E/libdetexhip_exp_ab_base.so:0,$E/libdetexhip_exp_ab_base.so:6,$E/libdetexhip_exp_ab_base.so:7,$E/libdetexhip_exp_ab_prio1.so:6,$E/libdetexhip_exp_ab_prio1.so:7 --formats BPTC --streams U,C --rounds 2 --out $OUT/bc7_persistent.jsonl 2>>$OUT/ab.err | cut -c1-200
echo "== BC7 block-major"
timeout 600 python tools/gpu_ab.py --libs $PWD/detex_amd/lib/libdetexhip.so,$E/libdetexhip_r02.so,$E/libdetexhip_exp_plain.so,$E/libdetexhip_exp_prio1.so --formats BPTC --streams U,C --layout tiled --rounds 2 --out $OUT/bc7_tiled.jsonl 2>>$OUT/ab.err | cut -c1-200
echo "== BC7 16384^2 (fixed vs proportional cost)"
timeout 600 python tools/gpu_ab.py --libs $PWD/detex_amd/lib/libdetexhip.so,$E/libdetexhip_exp_plain.so,$E/libdetexhip_exp_prio1.so --formats BPTC --streams U --size 16384 --rounds 2 --out $OUT/bc7_16384.jsonl 2>>$OUT/ab.err | cut -c1-200
echo "== clocks: product vs decode-only vs store-only"
timeout 600 python tools/gpu_ab.py --libs $PWD/detex_amd/lib/libdetexhip.so,$E/libdetexhip_exp_nostore.so,$E/libdetexhip_exp_nocompute.so --formats BPTC,BPTC_SIGNED_FLOAT,BC1 --streams U --rounds 1 --clocks --out $OUT/clocks.jsonl 2>>$OUT/ab.err | cut -c1-330
echo "== BC6H"
timeout 600 python tools/gpu_ab.py --libs $PWD/detex_amd/lib/libdetexhip.so,$E/libdetexhip_r02.so,$E/libdetexhip_exp_bc6prio1.so --formats BPTC_FLOAT,BPTC_SIGNED_FLOAT --streams U,C --rounds 2 --out $OUT/bc6h_ab.jsonl 2>>$OUT/ab.err | cut -c1-200
echo "== BC6H beyond the Infinity Cache: 16384x8192, U vs C"
timeout 600 python tools/gpu_ab.py --libs $PWD/detex_amd/lib/libdetexhip.so --formats BPTC_FLOAT --streams U,C --size 16384 --height 8192 --rounds 2 --out $OUT/bc6h_16384x8192.jsonl 2>>$OUT/ab.err | cut -c1-200
tail -5 $OUT/ab.err
echo "== done"
