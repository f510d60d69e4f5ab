#!/bin/bash
# The driver's multi-GPU commands, run by hand with RCCL's own log switched on:   bash tools/scale_preflight.sh [N ...]   (default: 2 4 8)
#   python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W
# with NCCL_DEBUG=INFO; one JSON line and one RCCL log per N under gpurun_out/scale_preflight/.  On a box with fewer than N GPUs bench.py
# ends with exit code 5 BEFORE any RCCL call (its rank census over gloo finds two ranks on one GPU) and this script stops there with the
# same code: that is the expected outcome on the one-GPU boxes of the build rounds (tests/test_gpu_rccl_preflight.py checks exactly that).
# N = 1 runs the same branch with DETEX_BENCH_FORCE_DIST=1: the RCCL path at world size 1 (process group, census, barriers, whole-image
# digests, both gathers), which a one-GPU box CAN run.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/scale_preflight; mkdir -p $OUT
STEPS=${STEPS:-20}; WARMUP=${WARMUP:-5}; PORT=${PORT:-29611}
export HSA_ENABLE_IPC_MODE_LEGACY=0 NCCL_DEBUG=${NCCL_DEBUG:-INFO} NCCL_DEBUG_SUBSYS=${NCCL_DEBUG_SUBSYS:-INIT,ENV,GRAPH}
NS=${*:-2 4 8}
for N in $NS; do
  echo "== N=$N"
  export NCCL_DEBUG_FILE=$PWD/$OUT/rccl_n${N}_%p.log       # (RCCL's INFO lines would otherwise go to stdout, into the JSON line's file)
  if [ "$N" = 1 ]; then
    DETEX_BENCH_FORCE_DIST=1 MASTER_PORT=$PORT timeout 900 python bench.py --gpus 1 --steps $STEPS --warmup $WARMUP --no-cpu > $OUT/bench_n1_forced.json 2> $OUT/rccl_n1.log
    rc=$?
  else
    timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((PORT + N)) bench.py --gpus $N --steps $STEPS --warmup $WARMUP \
      > $OUT/bench_n$N.json 2> $OUT/rccl_n$N.log
    rc=$?
  fi
  f=$OUT/bench_n$N.json; [ "$N" = 1 ] && f=$OUT/bench_n1_forced.json
  echo "exit code $rc"; grep -m3 "NOT one rank per GPU" $OUT/rccl_n$N.log | cut -c1-300; tail -1 $f | cut -c1-400
  cat $OUT/rccl_n${N}_*.log 2>/dev/null | grep -m1 -i "NCCL version\|RCCL version"
  # (torch.distributed.run reports a failed worker with its own exit code 1: bench.py's code 5 is recognised by its message)
  if [ $rc -ne 0 ] && grep -q "NOT one rank per GPU" $OUT/rccl_n$N.log; then echo "stopping: this box has fewer than $N GPUs (bench.py exit code 5, no RCCL call was made)"; exit 5; fi
  if [ $rc -ne 0 ]; then echo "stopping: N=$N ended with exit code $rc"; tail -5 $OUT/rccl_n$N.log | cut -c1-300; exit $rc; fi
done
