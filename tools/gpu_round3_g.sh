#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r03g
rm -rf $OUT; mkdir -p $OUT
E=build/explib
echo "== BC6H U vs C: product / decode-only / store-only"
timeout 600 python tools/gpu_ab.py --libs $PWD/detex_amd/lib/libdetexhip.so,$E/libdetexhip_exp_nostore.so,$E/libdetexhip_exp_nocompute.so --formats BPTC_FLOAT --streams U,C --rounds 2 --out $OUT/bc6h_split.jsonl 2>>$OUT/ab.err | cut -c1-200
echo "== write-path counters: BC6H U vs C beyond the Infinity Cache, BC1 at aligned / padded / odd pitches, the reference fill at the same pitches"
CASES=("BPTC_FLOAT U 16384 8192" "BPTC_FLOAT C 16384 8192" "BC1 U 8192 8192" "BC1 U 8192 8192 128" "BC1 U 8192 8192 16" "BC1 U 8190 8190" "BC1 U 8192 8192 0 8 linear fill" "BC1 U 8192 8192 128 8 linear fill" "BC1 U 8192 8192 16 8 linear fill")
bash tools/gpu_pmc_cases.sh $OUT "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" "${CASES[@]}" 2>&1 | cut -c1-300
bash tools/gpu_pmc_cases.sh $OUT "TCC_EA0_WRREQ_WRITE_DRAM_32B_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WR_UNCACHED_32B_sum TCC_EA0_WRREQ_LEVEL_sum" "${CASES[@]}" 2>&1 | cut -c1-300
bash tools/gpu_pmc_cases.sh $OUT "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" "BPTC_FLOAT U 8192 8192" "BPTC_FLOAT C 8192 8192" "BPTC U 8192 8192" "BPTC M 8192 8192" "BPTC_SIGNED_FLOAT U 8192 8192" 2>&1 | cut -c1-400
echo "== done"
