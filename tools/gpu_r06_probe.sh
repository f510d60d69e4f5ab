set -u
mkdir -p gpurun_out/r06
python bench.py --steps 20 --warmup 5 > gpurun_out/r06/bench_a.json 2> gpurun_out/r06/bench_a.err; tail -3 gpurun_out/r06/bench_a.err
python - <<'PY'
import json
r = json.load(open("gpurun_out/r06/bench_a.json"))
print(json.dumps({k: r[k] for k in ("value", "ms_per_step", "roofline")})[:600])
print(json.dumps(r.get("strong_image_32768"), indent=1)); print(json.dumps(r.get("bc6h_32768_whole"), indent=1))
print(json.dumps(r["host_tier_small"].get("oneshot_compiled_c_client"), indent=1)); print(json.dumps(r["host_tier_small"].get("oneshot_reference_compiled_c_client")))
PY
