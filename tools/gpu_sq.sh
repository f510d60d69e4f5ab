#!/bin/bash
# SQ counter pass for one format: bash tools/gpu_sq.sh BPTC_FLOAT
set -u
export TMPDIR=/tmp
FMT=$1; OUT=gpurun_out; mkdir -p $OUT; ROOT=$(pwd)
cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace -T -d $ROOT/$OUT/prof_sq_$FMT -o sq --output-format csv -- python $ROOT/bench.py --format $FMT --steps 10 --warmup 2 --no-cpu > $ROOT/$OUT/prof_sq_$FMT.log 2>&1
cd $ROOT; f=$(find $OUT/prof_sq_$FMT -name "*counter_collection.csv" | head -1); [ -n "$f" ] && grep decode_linear "$f" | head -8 | cut -d, -f9,13,16,17
cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --kernel-trace -T -d $ROOT/$OUT/prof_sq2_$FMT -o sq --output-format csv -- python $ROOT/bench.py --format $FMT --steps 10 --warmup 2 --no-cpu > $ROOT/$OUT/prof_sq2_$FMT.log 2>&1
cd $ROOT; f=$(find $OUT/prof_sq2_$FMT -name "*counter_collection.csv" | head -1); [ -n "$f" ] && grep decode_linear "$f" | head -7 | cut -d, -f9,16,17
tail -3 $OUT/prof_sq2_$FMT.log
