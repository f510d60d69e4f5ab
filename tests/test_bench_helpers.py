"""bench.py's helpers that decide what the driver's line may claim, exercised without a GPU: the telemetry sampler on a fake hwmon
directory, the live PMC pass declining (instead of failing) where it cannot run, and the per-rank band digests' bookkeeping."""
import importlib.util
import json
import os
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_telemetry_samples_a_hwmon_directory(bench, tmp_path):
    (tmp_path / "freq1_input").write_text("1850000000\n")
    (tmp_path / "power1_input").write_text("1399000000\n")
    t = bench.Telemetry(None, 0, hwmon_dir=str(tmp_path))
    calls = []
    row = t.during(lambda: (calls.append(1), time.sleep(0.001)), seconds=0.05)
    assert row["sclk_mhz"] == 1850 and row["power_w"] == 1399 and row["samples"] >= 1 and len(calls) >= 5
    (tmp_path / "power1_input").unlink()                       # boards without a power sensor: the clock alone
    row = t.during(lambda: time.sleep(0.001), seconds=0.03)
    assert row["sclk_mhz"] == 1850 and "power_w" not in row


def test_telemetry_without_sensors_reports_nothing(bench, tmp_path):
    t = bench.Telemetry(None, 0, hwmon_dir=str(tmp_path / "absent"))
    assert t.during(lambda: None, seconds=0.02) is None


def test_live_pmc_pass_declines_where_it_cannot_run(bench, monkeypatch):
    import shutil
    monkeypatch.setattr(shutil, "which", lambda name: None)
    assert bench.live_pmc_traffic("BC1", 8192) is None                     # no rocprofv3: the line falls back to the replayed value
    monkeypatch.setattr(shutil, "which", lambda name: "/opt/rocm/bin/rocprofv3")
    monkeypatch.setenv("ROCPROFILER_REGISTER_FORCE_LOAD", "1")
    assert bench.live_pmc_traffic("BC1", 8192) is None                     # bench.py itself under a profiler: no nested pass


def test_band_goldens_cover_every_world_size_the_bench_checks():
    """bench.py --gpus N digests each rank's band as 8 / N consecutive eighths of the golden image: the eighths exist for both sharded
    formats, are equal-sized and tile the image"""
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "digests_8192.json")))["bands_all"]
    for name, total in (("BC1", 32768 * 32768 * 4), ("BPTC_FLOAT", 32768 * 32768 * 8)):
        rows = [gold["%s/32768/%dof8" % (name, e)] for e in range(8)]
        assert sum(r["bytes"] for r in rows) == total and len({r["bytes"] for r in rows}) == 1
        assert [r["row0"] for r in rows] == [1024 * e for e in range(8)] and rows[-1]["row1"] == 8192
    for g in range(4):                                                   # a quarter band (north_star: BC1 over 4 GPUs) = two eighths
        q = gold["BC1/32768/%dof4" % g]
        assert q["bytes"] == 2 * gold["BC1/32768/0of8"]["bytes"] and q["row0"] == gold["BC1/32768/%dof8" % (2 * g)]["row0"]


def test_roofline_rows_claim_no_fraction_where_it_would_not_be_an_hbm_fraction(bench):
    """`frac` is null for a footprint that fits the 256 MiB Infinity Cache and for a rate above the 8 TB/s peak (no
    fraction above 1 anywhere); `write_frac` is given wherever the footprint exceeds the cache"""
    blocks = 2048 * 2048
    bc1 = bench.roofline_row(blocks * 72, blocks * 64, 8192 * 8192, 41.3)                  # 288 MiB: beyond the cache
    assert bc1["cache_resident"] is False and abs(bc1["frac"] - 0.914) < 0.002 and abs(bc1["write_frac"] - 0.8125) < 0.002 and "frac_note" not in bc1
    rgtc1 = bench.roofline_row(blocks * 24, blocks * 16, 8192 * 8192, 11.9)                # 96 MiB, 8.5 TB/s: cache-resident
    assert rgtc1["cache_resident"] is True and rgtc1["frac"] is None and rgtc1["write_frac"] is None and rgtc1["achieved_GBps"] > 8000
    pt = bench.roofline_row(4 * blocks * 72, 4 * blocks * 64, 16384 * 16384, 146.4)        # 1152 MiB at 8.25 TB/s: above the peak
    assert pt["cache_resident"] is False and pt["frac"] is None and 0.9 < pt["write_frac"] < 1.0 and "above the 8 TB/s peak" in pt["frac_note"]
    for launch_us in (10.0, 40.0, 160.0, 640.0):
        for scale in (1, 4):
            row = bench.roofline_row(scale * blocks * 80, scale * blocks * 64, scale * 8192 * 8192, launch_us)
            assert row["frac"] is None or row["frac"] <= 1.0


def test_bench_and_tools_have_no_undefined_names():
    """bench.py cannot run without a GPU, and this image has no linter: tools/check_names.py walks every function of it (and of the GPU-side
    measurement scripts) and reports names that are neither bound in an enclosing scope nor builtins -- what a typo in a rarely taken branch
    of the driver's command would otherwise turn into a NameError on the GPU box"""
    import subprocess, sys
    files = ["bench.py", "__graft_entry__.py", "tools/gpu_rotating.py", "tools/gpu_run_case.py", "tools/gpu_time.py", "tools/pmc_traffic.py"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_names.py")] + [os.path.join(ROOT, f) for f in files], capture_output=True, text=True)
    problems = [l for l in r.stdout.splitlines() if "undefined name __file__" not in l]
    assert not problems, problems
