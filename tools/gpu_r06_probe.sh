# round-6 probe (scratch): read-ahead A/B on the real kernels, one-shot client under the HIP API trace
set -u
export TMPDIR=/tmp
ROOT=$(pwd); O=$ROOT/gpurun_out/r06/footprint5; mkdir -p $O
B=$ROOT/tools/ubench/big_footprint
python -m pytest tests/test_ab_variants.py tests/test_gpu_host_multi.py -q -m gpu 2>&1 | tail -3
$B sweep BC1 24576 32768 2>&1 | tee $O/sweep_bc1.jsonl
$B sweep BPTC_FLOAT 16384 32768 2>&1 | tee $O/sweep_bc6h.jsonl
for RA in 1; do
    d=$O/pmc_ra$RA
    (cd /tmp && DETEXHIP_READ_AHEAD=$RA timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum --kernel-trace -d $d -o p --output-format csv -- $B pmc BC1 32768 > $d.log 2>&1)
    f2=$(find $d -name "*kernel_trace.csv" | head -1)
    python3 - "$f2" <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])): d[r["Kernel_Name"][:50]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000)
for k, v in d.items(): print("   ", k, "median us", sorted(v)[len(v)//2], "n", len(v))
PY
    f=$(find $d -name "*counter_collection.csv" | head -1)
    python3 - "$f" $RA <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows: d[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in d.items(): print("read_ahead", sys.argv[2], k, {c: sorted(x)[len(x)//2] for c, x in v.items()}, len(list(v.values())[0]))
PY
    rm -rf $d
done 2>&1 | tee $O/pmc_read_ahead.txt
V="BC1 BC1A BC2 BC3 RGTC1 RGTC2 SIGNED_RGTC1 SIGNED_RGTC2 BPTC BPTC_FLOAT ETC1 ETC2 ETC2_PUNCHTHROUGH ETC2_EAC EAC_R11 EAC_RG11 EAC_SIGNED_R11"; F=""; for v in $V; do F="$F $ROOT/tests/golden/test-texture-$v.ktx"; done
python3 - $F <<'PY' 2>&1 | tee $O/oneshot_wall.txt
import subprocess, sys, time, os
root = os.getcwd()
for exe in ("detex_client", "detex_client_reflib"):
    ts = []
    for k in range(6):
        t0 = time.perf_counter(); out = subprocess.run([os.path.join(root, "tests/c_client", exe), "--oneshot"] + sys.argv[1:], capture_output=True, text=True).stdout; ts.append((time.perf_counter() - t0) * 1e3)
    print(exe, "process wall ms", ["%.1f" % t for t in ts], out.strip()[:160])
t = []
for k in range(4):
    t0 = time.perf_counter(); subprocess.run([os.path.join(root, "tools/ubench/host_latency"), "none"], capture_output=True); t.append((time.perf_counter() - t0) * 1e3)
print("host_latency none (a HIP program that initialises and exits?) wall ms", ["%.1f" % x for x in t])
PY
d=$O/hiptrace
(cd /tmp && timeout 300 rocprofv3 --hip-trace --kernel-trace --stats -d $d -o t --output-format csv -- $ROOT/tests/c_client/detex_client --oneshot $F > $d.log 2>&1)
ls $d/* | head; f=$(find $d -name "*hip_api_stats.csv" | head -1); [ -n "$f" ] && head -30 $f | tee $O/oneshot_hip_api_stats.csv
f=$(find $d -name "*hip_api_trace.csv" | head -1); [ -n "$f" ] && cp $f $O/oneshot_hip_api_trace.csv
rm -rf $d
