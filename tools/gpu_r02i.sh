#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02i; mkdir -p $OUT; ROOT=$(pwd)
echo "== N=2 code path over gloo (two ranks share the one GPU: plumbing check, not a measurement)"
DETEX_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 3 > $OUT/bench_n2_gloo.json 2> $OUT/bench_n2.err; cut -c1-2500 $OUT/bench_n2_gloo.json; tail -5 $OUT/bench_n2.err
echo "== N=2 --weak"
DETEX_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 10 --warmup 3 --weak > $OUT/bench_n2_weak.json 2>> $OUT/bench_n2.err; cut -c1-600 $OUT/bench_n2_weak.json
echo "== single-GPU RCCL (world 1 nccl init is skipped); strong image on one GPU"
timeout 600 python bench.py --strong-image 32768 --steps 10 --warmup 3 --no-cpu --no-extras 2>>$OUT/bench_n2.err | cut -c1-700
echo "== done"
