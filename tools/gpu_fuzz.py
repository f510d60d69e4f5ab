#!/usr/bin/env python3
"""Randomised parity sweep on the GPU box: seeds x formats x geometries, linear + block-major + per-block API against the
oracle (test infrastructure).  usage: python tools/gpu_fuzz.py [seconds=60] [seed0=1]
Geometries include widths on both sides of the several-blocks-per-lane condition and clipped sizes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from detex_amd import binding, formats as F
import oracle_lib as ol

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
oracle = ol.Oracle()
binding.load()
t0 = time.time(); cases = 0
geoms = [(256, 64), (1024, 128), (72, 40), (100, 36), (4, 4), (260, 12), (1028, 8), (62, 30), (513, 17)]
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    for fmt in F.FORMATS:
        W, H = geoms[int(rng.integers(0, len(geoms)))]
        wb, hb = (W + 3) // 4, (H + 3) // 4
        data = ol.stream_u(fmt, wb * hb, seed=int(rng.integers(1, 1 << 40)))
        # bias some streams towards rare modes: copy a random block over a few others (uniform waves, repeated planar blocks)
        if rng.integers(0, 3) == 0:
            blk = data.reshape(-1, fmt.block_bytes)
            src = blk[int(rng.integers(0, len(blk)))].copy()
            blk[rng.integers(0, len(blk), max(1, len(blk) // int(rng.integers(2, 40))))] = src
        dev = torch.from_numpy(np.ascontiguousarray(data)).cuda()
        ok_o, want = oracle.linear(fmt, data, W, H)
        status = torch.zeros(1, dtype=torch.int32, device="cuda")
        got = binding.decompress_linear_device(fmt, dev, W, H, status=status)
        torch.cuda.synchronize()
        assert np.array_equal(got.cpu().numpy().reshape(-1), np.asarray(want).reshape(-1)), ("linear", fmt.name, W, H, seed)
        assert bool(status.item() == 0) == ok_o, ("status", fmt.name, W, H, seed)
        ok_t, want_t = oracle.tiled(fmt, data, wb, hb)
        got_t = binding.decompress_tiled_device(fmt, dev, wb, hb)
        torch.cuda.synchronize()
        assert np.array_equal(got_t.cpu().numpy(), want_t), ("tiled", fmt.name, W, H, seed)
        mask = int(rng.integers(0, 1 << 14)) | (0 if rng.integers(0, 2) else 0xFFFFFFFF)
        ok_b, want_b = oracle.blocks(fmt, data, mode_mask=mask)
        got_b, got_ok = binding.decompress_blocks_device(fmt, dev, wb * hb, mode_mask=mask)
        torch.cuda.synchronize()
        assert np.array_equal(got_ok.cpu().numpy()[:wb * hb].astype(bool), ok_b), ("blocks ok", fmt.name, seed, hex(mask))
        assert np.array_equal(got_b.cpu().numpy().reshape(-1)[:wb * hb * 16 * fmt.pixel_bytes], want_b.reshape(-1)), ("blocks", fmt.name, seed, hex(mask))
        cases += 3
    seed += 1
print("fuzz: %d cases, seeds up to %d, %.0f s: all bit-exact" % (cases, seed - 1, time.time() - t0))
