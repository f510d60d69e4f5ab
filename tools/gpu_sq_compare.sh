#!/bin/bash
export TMPDIR=/tmp
ROOT=$(pwd); OUT=gpurun_out/sqc; rm -rf $OUT; mkdir -p $OUT
for lib in libdetexhip libdetexhip_r01; do
 cd /tmp && DETEXHIP_LIB=$ROOT/detex_amd/lib/$lib.so timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --kernel-trace -T -d $ROOT/$OUT/p_$lib -o sq --output-format csv -- python $ROOT/bench.py --format BPTC --stream C --steps 10 --warmup 2 --no-cpu --no-extras > $ROOT/$OUT/$lib.log 2>&1
 cd $ROOT; f=$(find $OUT/p_$lib -name "*counter_collection.csv" | head -1); echo "== $lib"; [ -n "$f" ] && grep decode_linear "$f" | head -8 | cut -d, -f13,16,17
 cd /tmp && DETEXHIP_LIB=$ROOT/detex_amd/lib/$lib.so timeout 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace -T -d $ROOT/$OUT/q_$lib -o sq --output-format csv -- python $ROOT/bench.py --format BPTC --stream C --steps 10 --warmup 2 --no-cpu --no-extras > $ROOT/$OUT/${lib}_lds.log 2>&1
 cd $ROOT; f=$(find $OUT/q_$lib -name "*counter_collection.csv" | head -1); [ -n "$f" ] && grep decode_linear "$f" | head -6 | cut -d, -f13,16,17
done
