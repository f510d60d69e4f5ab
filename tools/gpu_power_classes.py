#!/usr/bin/env python3
"""Board power and shader clock while ONE instruction class runs on every SIMD (tools/ubench/valu_rates loop NAME SECONDS): what a
wave64 instruction of each class costs in energy, for kernels that run at the board's power cap (BC7, signed BC6H on random data --
there, time = energy / cap, so joules per instruction price a kernel better than issue cycles do).
usage (GPU box): python tools/gpu_power_classes.py [seconds=1.5] [class ...]"""
import json, os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import Telemetry

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5
classes = sys.argv[2:] or ["v_add_u32", "v_and_or_b32", "v_bitop3", "v_lshrrev_b32", "v_lshlrev", "v_bfe_u32", "v_perm_b32", "v_mad_u32_u24", "v_mul_u32_u24",
                           "v_pk_mad_u16", "v_pk_add_u16", "pk_mul_lo", "and_sdwa", "v_cndmask_b32", "v_mov", "ds_read_b128"]
tel = Telemetry(torch, 0)
exe = os.path.join(ROOT, "tools", "ubench", "valu_rates")


def sample(duration):
    out = []
    t_end = time.perf_counter() + duration
    while time.perf_counter() < t_end:
        out.append((tel._read("freq1_input"), tel._read("power1_input")))
        time.sleep(0.002)
    return out


def summary(samples):
    s = [x for x in samples[len(samples) // 2:] if x[0] and x[1]]
    if not s:
        return None, None
    f = sorted(x[0] for x in s)[len(s) // 2] / 1e6
    p = sorted(x[1] for x in s)[len(s) // 2] / 1e6
    return round(f), round(p)


time.sleep(1.0)
idle_f, idle_p = summary(sample(0.5))
print(json.dumps({"class": "(idle)", "sclk_mhz": idle_f, "power_w": idle_p}), flush=True)
for name in classes:
    proc = subprocess.Popen([exe, "loop", name, str(seconds)], stdout=subprocess.PIPE, text=True)
    time.sleep(0.3)                                   # (process start-up, code object load)
    got = sample(seconds - 0.5)
    line = proc.communicate()[0].strip()
    f, p = summary(got)
    row = {"class": name, "sclk_mhz": f, "power_w": p}
    try:
        rate = float(line.split()[-1])
        row["wave_instructions_per_s"] = rate
        if p and idle_p:
            row["nJ_per_wave_instruction"] = round((p - idle_p) / rate * 1e9, 3)
            row["cycles_per_instruction_per_simd"] = round(f * 1e6 * 1024 / rate, 2) if f else None
    except Exception:  # noqa
        row["output"] = line
    print(json.dumps(row), flush=True)
    time.sleep(0.5)
