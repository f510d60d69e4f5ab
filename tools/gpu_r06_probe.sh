set -u
mkdir -p gpurun_out/r06
python bench.py --steps 20 --warmup 5 > gpurun_out/r06/bench_b.json 2> gpurun_out/r06/bench_b.err; tail -3 gpurun_out/r06/bench_b.err
python - <<'PY'
import json
r = json.load(open("gpurun_out/r06/bench_b.json"))
print(r["value"], r["roofline"]["frac"], r["roofline"]["launch_us"])
print(json.dumps({k: v for k, v in r["roofline"]["blocks_from_hbm"].items() if k != "note"}))
for k, v in r["per_format"]["formats"].items():
    if "blocks_from_hbm" in v: print(k, v["launch_us"], v["frac"], json.dumps(v["blocks_from_hbm"]))
PY
