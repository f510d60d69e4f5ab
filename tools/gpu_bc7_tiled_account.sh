#!/bin/bash
# Block-major BC7 against linear BC7, same run: per-wave instruction counters of decode_blocks<DecBPTC, 0, false> and
# decode_linear<DecBPTC, 0, true> on stream U at 8192^2, launch times with clock / power samples.   bash tools/gpu_bc7_tiled_account.sh [OUTDIR]
set -u
export TMPDIR=/tmp
ROOT=$(pwd); OUT=${1:-gpurun_out/r06/bc7_tiled}; mkdir -p $OUT
rm -f $OUT/counters.jsonl
for group in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  bash tools/gpu_pmc_cases.sh $OUT "$group" "BPTC U 8192 8192 0 8 linear" "BPTC U 8192 8192 0 8 tiled" "BC1 U 8192 8192 0 8 linear" "BC1 U 8192 8192 0 8 tiled"
done
for rep in 1 2; do
  python tools/gpu_time.py BPTC U linear 8192 linear_$rep; python tools/gpu_time.py BPTC U tiled 8192 tiled_$rep
done | tee $OUT/times.jsonl
python tools/gpu_ab.py --libs detex_amd/lib/libdetexhip.so --formats BPTC --streams U --clocks 2>&1 | tail -4 | tee $OUT/clocks_linear.txt
python tools/gpu_ab.py --libs detex_amd/lib/libdetexhip.so --formats BPTC --streams U --layout tiled --clocks 2>&1 | tail -4 | tee $OUT/clocks_tiled.txt
