#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r03f
rm -rf $OUT; mkdir -p $OUT
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3 | tee $OUT/pytest_gpu.log
echo "== bench N=1"; /usr/bin/time -v timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -25 $OUT/bench.err | grep -E "Elapsed|Maximum resident|rror|failed"; cat $OUT/bench.json
echo "== bench N=2 over gloo on one GPU (plumbing)"
DETEX_BENCH_BACKEND=gloo timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 2 --steps 5 --warmup 2 > $OUT/bench_n2_gloo_one_gpu.json 2> $OUT/bench_n2.err; tail -3 $OUT/bench_n2.err; cat $OUT/bench_n2_gloo_one_gpu.json
echo "== small latency"; python tools/gpu_small_latency.py detex_amd/lib/libdetexhip.so 2>/dev/null | tee $OUT/small_latency.jsonl
echo "== done"
