#!/bin/bash
# The final measurement round of a build round on the GPU box (via gpurun):  bash tools/gpu_evidence_round.sh r05  -> gpurun_out/r05/, whose
# summaries are then committed under profiles/r05/ (what profiles/rNN/ holds from the FINAL library of a round comes from here).
# SKIP="fuzz pmc" leaves sections out.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-round}
OUT=gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
ROOT=$(pwd)
skip() { case " ${SKIP:-} " in *" $1 "*) return 0 ;; esac; return 1; }
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -s 2>&1 | tail -40 | tee $OUT/pytest_gpu.log | tail -3
mkdir -p $OUT/rccl_preflight; cp gpurun_out/rccl_preflight/*.log gpurun_out/rccl_preflight/*.json $OUT/rccl_preflight/ 2>/dev/null
echo "== bench (driver line, N=1)"; ( time timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real; cut -c1-400 $OUT/bench.json
echo "== compiled C client: fixtures, latency, batched blocks"
tests/c_client/detex_client tests/golden/test-texture-BC1.ktx tests/golden/test-texture-BPTC_FLOAT.ktx | tee $OUT/c_client.txt
tests/c_client/detex_client --latency | tee -a $OUT/c_client.txt | head -3
echo "-- pixel buffers from detexhipAllocPixelBuffer (pinned: the kernel writes straight into them)" >> $OUT/c_client.txt; tests/c_client/detex_client --latency owned >> $OUT/c_client.txt
echo "-- the same with DETEXHIP_RESIDENT_US=0 (a launch per call)" >> $OUT/c_client.txt; DETEXHIP_RESIDENT_US=0 tests/c_client/detex_client --latency >> $OUT/c_client.txt
echo "-- the same program linked against the compiled reference (one host thread)" >> $OUT/c_client.txt; [ -x tests/c_client/detex_client_reflib ] && tests/c_client/detex_client_reflib --latency >> $OUT/c_client.txt
echo "-- n independent blocks: the loop over the leaf function against ONE detexhipDecompressBlocks call" >> $OUT/c_client.txt; tests/c_client/detex_client --blocks | tee -a $OUT/c_client.txt
echo "-- the same loop in the compiled reference" >> $OUT/c_client.txt; [ -x tests/c_client/detex_client_reflib ] && tests/c_client/detex_client_reflib --blocks >> $OUT/c_client.txt
ldd tests/c_client/detex_client | grep -i "amdhip\|detexhip" >> $OUT/c_client.txt
echo "== host-tier test program with the device (uninstrumented: the pool runs no sanitizer builds)"; timeout 300 tests/host_san/api_plain 2>&1 | tail -3 | tee $OUT/api_plain_gpu.txt
echo "== host tier: where a mid-size call's time goes; the runtime's pageable copy curve"; timeout 300 tools/ubench/host_midsize > $OUT/host_midsize.jsonl 2>&1; wc -l $OUT/host_midsize.jsonl
echo "== cold start"; timeout 300 python tools/gpu_cold_trace.py BC1 > $OUT/cold_trace_bc1.json 2> /dev/null; cut -c1-200 $OUT/cold_trace_bc1.json
echo "== per-format tables: all formats, streams U / M / C, linear and block-major at 8192^2; linear at 16384^2 (beyond the Infinity Cache for every format)"
timeout 900 python bench.py --no-cpu --no-extras --formats-json $OUT/formats_8192.json > /dev/null 2> $OUT/formats.err; grep -c launch_us $OUT/formats.err
timeout 900 python bench.py --no-cpu --no-extras --layout tiled --formats-json $OUT/formats_8192_tiled.json > /dev/null 2> $OUT/formats_tiled.err; grep -c launch_us $OUT/formats_tiled.err
timeout 1200 python bench.py --no-cpu --no-extras --size 16384 --formats-json $OUT/formats_16384.json > /dev/null 2> $OUT/formats_16384.err; grep -c launch_us $OUT/formats_16384.err
echo "== N=2 plumbing over gloo on this one GPU: the 32768^2 BC1 image in two bands, every rank digests its band (eighths) against the reference"
DETEX_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29617 bench.py --gpus 2 --strong-image 32768 --steps 3 --warmup 1 --no-extras --no-cpu > $OUT/bench_n2_gloo_32768.json 2> $OUT/bench_n2_gloo.err; cut -c1-300 $OUT/bench_n2_gloo_32768.json; tail -2 $OUT/bench_n2_gloo.err
echo "== the multi-GPU branch under RCCL at world size 1 (DETEX_BENCH_FORCE_DIST=1), RCCL's own log kept; then the driver's N=2 command, which must end with exit code 5 here"
STEPS=20 WARMUP=5 bash tools/scale_preflight.sh 1 2>&1 | tail -4; bash tools/scale_preflight.sh 2 > $OUT/scale_preflight_n2.txt 2>&1; echo "scale_preflight.sh 2 -> exit code $?" | tee -a $OUT/scale_preflight_n2.txt
mkdir -p $OUT/scale_preflight; cp gpurun_out/scale_preflight/bench_n1_forced.json $OUT/scale_preflight/ 2>/dev/null
for f in gpurun_out/scale_preflight/rccl_n1_*.log; do [ -f "$f" ] && grep -v "Channel [0-9]*/[0-9]* :" "$f" | cut -c1-400 | head -150 > $OUT/scale_preflight/$(basename $f); done
grep -m2 "NOT one rank per GPU" gpurun_out/scale_preflight/rccl_n2.log | cut -c1-400 >> $OUT/scale_preflight_n2.txt
echo "== rocprofv3 kernel trace + stats: the headline workload alone, and the driver's exact command"
bash tools/gpu_kernel_stats.sh 2>&1 | tail -4; cp gpurun_out/kernel_stats/headline_kernel_stats.csv $OUT/bc1_8192_kernel_stats.csv; cp gpurun_out/kernel_stats/driver_kernel_stats.csv $OUT/driver_command_kernel_stats.csv
cp gpurun_out/kernel_stats/headline_bench.json $OUT/bench_headline_under_rocprofv3.json
bash tools/gpu_rocprof_formats.sh 2>&1 | grep last200 | cut -c1-200; mkdir -p $OUT/rocprof_formats; cp gpurun_out/rocprof_formats/*.json gpurun_out/rocprof_formats/*kernel_stats.csv $OUT/rocprof_formats/ 2>/dev/null
if ! skip pmc; then
  echo "== PMC traffic (separate passes): nine kernels at 8192^2, the five narrow formats at 16384^2"
  timeout 2400 python tools/pmc_traffic.py $OUT BC1:linear BC3:linear BPTC:linear BPTC_FLOAT:linear BPTC_SIGNED_FLOAT:linear ETC2:linear ETC2_EAC:linear BC1:tiled BPTC:tiled \
    RGTC1:linear:16384 RGTC2:linear:16384 SIGNED_RGTC1:linear:16384 EAC_R11:linear:16384 EAC_SIGNED_R11:linear:16384 2>&1 | tail -15
fi
echo "== SQ counters per wave (BC7, BC6H)"
for FMT in BPTC BPTC_SIGNED_FLOAT BPTC_FLOAT; do
  cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace -d $ROOT/$OUT/sq_$FMT -o sq --output-format csv -- python $ROOT/tools/gpu_run_case.py $FMT U 8192 8192 0 6 linear > /dev/null 2>&1
  cd $ROOT; f=$(find $OUT/sq_$FMT -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "$FMT" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "decode_linear" in r["Kernel_Name"]]
d = collections.defaultdict(list)
for r in rows: d[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sorted(v)[len(v)//2] for k, v in d.items()}
w = m.get("SQ_WAVES", 1)
print(sys.argv[2], "stream U, per wave:", {k: round(v / w, 1) for k, v in m.items() if k != "SQ_WAVES"})
PY
  rm -rf $OUT/sq_$FMT
done | tee $OUT/sq_per_wave.txt
echo "== round 6: the large-footprint cliff (sweep, mock, mix, counters), blocks out of HBM (rotating inputs), block-major BC7 against linear, the one-shot client"
bash tools/gpu_big_footprint.sh $OUT/footprint > $OUT/footprint.log 2>&1; tail -2 $OUT/footprint/sweep_bc1.jsonl | cut -c1-300
rm -f $OUT/footprint/*.err
python tools/gpu_rotating.py detex_amd/lib/libdetexhip.so BC1,BC1A,BC2,BC3,RGTC2,SIGNED_RGTC2,BPTC_FLOAT,BPTC_SIGNED_FLOAT,BPTC,ETC1,ETC2,ETC2_PUNCHTHROUGH,ETC2_EAC,EAC_RG11,EAC_SIGNED_RG11 8192 1,2 2>/dev/null > $OUT/rotating_inputs_8192.jsonl
python tools/gpu_rotating.py detex_amd/lib/libdetexhip.so BC1,BC3,BPTC,ETC2,ETC2_EAC,BPTC_FLOAT 8192 1,2 tiled 2>/dev/null > $OUT/rotating_inputs_8192_tiled.jsonl; wc -l $OUT/rotating_inputs_8192*.jsonl
bash tools/gpu_bc7_tiled_account.sh $OUT/bc7_tiled > $OUT/bc7_tiled.log 2>&1; rm -f $OUT/bc7_tiled/*.log; cat $OUT/bc7_tiled/times.jsonl | cut -c1-160
V="BC1 BC1A BC2 BC3 RGTC1 RGTC2 SIGNED_RGTC1 SIGNED_RGTC2 BPTC BPTC_FLOAT ETC1 ETC2 ETC2_PUNCHTHROUGH ETC2_EAC EAC_R11 EAC_RG11 EAC_SIGNED_R11"; F=""; for v in $V; do F="$F $ROOT/tests/golden/test-texture-$v.ktx"; done
mkdir -p $OUT/oneshot
for i in 1 2 3 4 5 6 7 8 9 10; do tests/c_client/detex_client --oneshot-breakdown $F; done > $OUT/oneshot/oneshot_breakdown.txt 2>&1
[ -x tests/c_client/detex_client_reflib ] && for i in 1 2 3; do tests/c_client/detex_client_reflib --oneshot $F; done >> $OUT/oneshot/oneshot_breakdown.txt 2>&1
(cd /tmp && timeout 300 rocprofv3 --hip-trace --kernel-trace --stats -d $ROOT/$OUT/oneshot/trace -o t --output-format csv -- $ROOT/tests/c_client/detex_client --oneshot $F > $ROOT/$OUT/oneshot/trace.log 2>&1)
f=$(find $OUT/oneshot/trace -name "*hip_api_stats.csv" | head -1); [ -n "$f" ] && head -12 $f | tee $OUT/oneshot/oneshot_hip_api_stats.csv | head -6
f=$(find $OUT/oneshot/trace -name "*hip_api_trace.csv" | head -1); [ -n "$f" ] && grep -v "__hipRegister" $f > $OUT/oneshot/oneshot_hip_api_trace.csv
rm -rf $OUT/oneshot/trace $OUT/oneshot/trace.log
echo "== round 6 (second session): who wrote the blocks (upload / device copy before each launch); large textures through the host-pointer entry (duplex staged path)"
python tools/gpu_fresh_blocks.py BC1,BC3,BPTC_FLOAT 8192 80 2>/dev/null > $OUT/fresh_blocks.jsonl; python tools/gpu_fresh_blocks.py BC1 16384 40 2>/dev/null >> $OUT/fresh_blocks.jsonl; wc -l $OUT/fresh_blocks.jsonl
python tools/gpu_host_big.py $ROOT/detex_amd/lib/libdetexhip.so 2>/dev/null > $OUT/host_big.jsonl; cut -c1-200 $OUT/host_big.jsonl | head -3
echo "== mode histograms / mip chains"; (timeout 300 python tools/bench_histogram.py 2>/dev/null) | tee $OUT/histogram.txt | cut -c1-120; timeout 300 python tools/bench_mips.py 2>/dev/null | tail -1 > $OUT/mips.json; cut -c1-200 $OUT/mips.json
if ! skip fuzz; then echo "== fuzz 150 s"; timeout 400 python tools/gpu_fuzz.py 150 50000 2>&1 | tail -1 | tee $OUT/fuzz.log; fi
rm -rf $OUT/pmc_*_*_*_* $OUT/pmc_*_*_* 2>/dev/null
echo "== done"
