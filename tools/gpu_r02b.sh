#!/bin/bash
# round-2 pass B: where does the time of the VALU-heavy kernels go?  decode-without-stores, stores-without-decode,
# occupancy sensitivity (LDS padding), LDS conflict counters, round-1 BC7 decoder, host transfer paths
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02b; mkdir -p $OUT; ROOT=$(pwd)
FM=BPTC,BPTC_SIGNED_FLOAT,BPTC_FLOAT,ETC2_EAC,ETC2,RGTC1,SIGNED_RGTC1,EAC_SIGNED_R11,BC1
for lib in libdetexhip libdetexhip_exp_nostore libdetexhip_exp_nocompute libdetexhip_exp_pad8192 libdetexhip_exp_pad16384; do
  DETEXHIP_LIB=$ROOT/detex_amd/lib/$lib.so timeout 300 python tools/gpu_time.py $FM U 2>>$OUT/err.log | tee -a $OUT/times.jsonl
done
echo "== BC7 round-1 decoder (A/B build variant 4), U M C"
DETEXHIP_LIB=$ROOT/detex_amd/lib/libdetexhip_ab.so DETEXHIP_VARIANT=4 timeout 300 python tools/gpu_time.py BPTC U,M,C linear 8192 r01_variant4 2>>$OUT/err.log | tee -a $OUT/times.jsonl
timeout 300 python tools/gpu_time.py BPTC U,M,C 2>>$OUT/err.log | tee -a $OUT/times.jsonl
echo "== A/B kernels parity"; DETEXHIP_LIB=$ROOT/detex_amd/lib/libdetexhip_ab.so timeout 300 python -m pytest tests/test_ab_variants.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3 | tee $OUT/pytest_ab.log
echo "== parity of the BPTC paths after the kernel-skeleton change"; timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "BPTC or digest or random or ragged" 2>&1 | tail -3
echo "== LDS counters BPTC"
cd /tmp && timeout 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace -T -d $ROOT/$OUT/prof_lds_BPTC -o sq --output-format csv -- python $ROOT/bench.py --format BPTC --steps 10 --warmup 2 --no-cpu --no-extras > $ROOT/$OUT/prof_lds_BPTC.log 2>&1
cd $ROOT; f=$(find $OUT/prof_lds_BPTC -name "*counter_collection.csv" | head -1); [ -n "$f" ] && grep decode_linear "$f" | head -8 | cut -d, -f9,16,17
echo "== host transfer paths"; timeout 120 ./tools/ubench/host_paths 2>&1 | tee $OUT/host_paths.txt
echo "== clocks under BPTC"; timeout 120 python tools/gpu_sustain.py BPTC 24 2>&1 | tail -2 | cut -c1-1500
echo "== done"
