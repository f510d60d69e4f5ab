"""Pre-flight of the multi-GPU path on ONE GPU: everything bench.py --gpus N and detex_amd/sharding.py do
between ranks, executed under the REAL collective library -- torch.distributed backend `nccl` = RCCL -- at world size 1, so that the first
contact with an 8-GPU node is not the first execution of this code.  (RCCL refuses two ranks on one GPU -- "Duplicate GPU detected" -- so
world size 1 plus a grouped send / receive to itself is what a one-GPU box can run.)  RCCL's own init log goes to gpurun_out/."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "rccl_preflight")
pytestmark = pytest.mark.gpu


def _env(port, log="rccl", **extra):
    # (RCCL prints its INFO lines to STDOUT unless told otherwise: NCCL_DEBUG_FILE keeps the JSON line of the worker / of bench.py clean)
    os.makedirs(OUT, exist_ok=True)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,ENV",
               NCCL_DEBUG_FILE=os.path.join(OUT, log + "_%p.log"))
    env.update(extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    return env


def _last_json(text):
    for line in reversed(text.strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise AssertionError("no JSON line in:\n" + text[-3000:])


def test_sharding_gathers_under_rccl_world_size_1():
    """sharding.gather_image / gather_image_to_root (default group and a sub-group) and a grouped isend / irecv pair to itself, CUDA tensors,
    backend nccl: every gathered image == the oracle"""
    os.makedirs(OUT, exist_ok=True)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_preflight_worker.py")], capture_output=True, text=True, timeout=600,
                       env=_env(29533 + os.getpid() % 200, log="rccl_sharding_worker"))
    open(os.path.join(OUT, "sharding_worker.out"), "w").write(r.stderr[-20000:] + "\n--- stdout\n" + r.stdout)
    assert r.returncode == 0, r.stderr[-3000:]
    res = _last_json(r.stdout)
    assert res["backend"] == "nccl" and res["ok"], res
    assert len(res["checks"]) == 1 + 3 * 2 * 2 + 1
    print("RCCL", res["rccl_version"], "checks:", len(res["checks"]))


def test_bench_multi_gpu_branch_under_rccl_world_size_1():
    """bench.py's whole N > 1 branch (DETEX_BENCH_FORCE_DIST=1, WORLD_SIZE 1, backend nccl): mixed-backend process group, rank census, the
    RCCL all_reduce, barriers over RCCL, the 32768^2 BC1 image decoded and digested WHOLE against the compiled reference's eighths, both
    gathers (timed), BASELINE configs[4] (BC6H 32768^2: 8 GiB of pixels) with its digests and gathers, the weak line"""
    os.makedirs(OUT, exist_ok=True)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu"], capture_output=True, text=True,
                       timeout=1200, env=_env(29733 + os.getpid() % 200, log="rccl_bench_forced_dist", DETEX_BENCH_FORCE_DIST="1"), cwd=ROOT)
    open(os.path.join(OUT, "bench_forced_dist.err"), "w").write(r.stderr[-20000:])
    open(os.path.join(OUT, "bench_forced_dist.json"), "w").write(r.stdout)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["forced_dist_path"] is True and d["collective_backend"] == "nccl" and d["rccl_ranks"] == 1 and d["n_gpus"] == 1
    assert d["scaling"] == "strong" and "32768x32768" in d["config"]["workload"] and d["value"] > 0
    assert d["whole_band_digests_match_reference_all_ranks"] is True
    for key in ("to_root", "to_all"):
        assert "error" not in d["gather"][key], d["gather"][key]
        assert d["gather"][key]["own_band_intact"] is True
    b6 = d["bc6h_32768"]
    assert "error" not in b6, b6
    assert b6["whole_band_digests_match_reference_all_ranks"] is True and b6["verified_bit_exact_rows_min_over_ranks"] > 0
    assert "error" not in b6["gather"]["to_root"] and "error" not in b6["gather"]["to_all"]
    assert d["weak"]["value_gpixel_s"] > 0
    hbm = d["blocks_from_hbm"]                                       # the headline's loop over different inputs per rank (like-for-like with the N = 1 whole image)
    assert "error" not in hbm and hbm["value_gpixel_s"] > 0 and hbm["inputs_per_rank"] >= 3, hbm
    print("bench.py N>1 branch under RCCL %s: value %.1f Gpixel/s, BC6H 32768^2 %.1f Gpixel/s" % (d.get("rccl_version"), d["value"], b6["value_gpixel_s"]))


def test_scale_preflight_script_ends_cleanly_without_enough_gpus():
    """tools/scale_preflight.sh = the driver's torchrun command with NCCL_DEBUG=INFO.  With fewer GPUs than ranks bench.py's census (over
    gloo, before any RCCL call) must end the run with its message and the script with exit code 5 -- not an RCCL abort; with enough GPUs
    the run must succeed"""
    import torch
    n = 2
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "scale_preflight.sh"), str(n)], capture_output=True, text=True, timeout=1500,
                       env=dict(_env(0, log="rccl_scale_script"), STEPS="3", WARMUP="1", PORT=str(29911 + os.getpid() % 50)), cwd=ROOT)
    print(r.stdout[-1500:])
    if torch.cuda.device_count() < n:
        assert r.returncode == 5, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
        assert "fewer than 2 GPUs" in r.stdout
        log = open(os.path.join(ROOT, "gpurun_out", "scale_preflight", "rccl_n2.log")).read()
        assert "NOT one rank per GPU" in log and "Duplicate GPU" not in log
    else:
        assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
        d = _last_json(open(os.path.join(ROOT, "gpurun_out", "scale_preflight", "bench_n2.json")).read())
        assert d["rccl_ranks"] == 2 and d["whole_band_digests_match_reference_all_ranks"] is True
