#!/usr/bin/env python3
"""Randomised parity sweep on the GPU box (tests/fuzz.py: seeds x formats x geometries, linear incl. padded pitches and
epilogue targets + block-major + per-block API with random mode masks, against the oracle).
usage: python tools/gpu_fuzz.py [seconds=60] [seed0=1]   (the `-m gpu` suite runs a fixed slice of the same cases)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from detex_amd import binding
import oracle_lib as ol
import fuzz

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
oracle = ol.Oracle()
binding.load()
t0 = time.time(); cases = 0
while time.time() - t0 < budget:
    cases += fuzz.run_seed(seed, oracle, binding, torch)
    seed += 1
print("fuzz: %d decode calls checked, seeds up to %d, %.0f s: all bit-exact" % (cases, seed - 1, time.time() - t0))
