#!/bin/bash
# same-run A/B of library builds: bash tools/gpu_ab3.sh "<formats>" "<streams>" lib1 lib2 ...   (results: gpurun_out/ab3.jsonl)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
FMTS=$1; STREAMS=$2; shift 2
for lib in "$@" "$1"; do
  DETEXHIP_LIB=$PWD/detex_amd/lib/$lib python tools/gpu_time.py "$FMTS" "$STREAMS" ${LAYOUT:-linear} 8192 $lib >> gpurun_out/ab3.jsonl 2>> gpurun_out/ab3.err
done
