#!/usr/bin/env python3
"""bench.py -- throughput of the block-decode hot path on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W   (N>1 via torch.distributed.run)
prints ONE JSON line on rank 0.

  step      one pass of the hot path over one batch: one detexhipDecompressTextureLinearDevice
            call (= one kernel launch) decoding a whole width x height block stream that is
            already resident in HBM into a device-resident linear image.
  workload  BASELINE.json configs[1]: BC1 -> RGBA8, 8192 x 8192, synthetic stream U
            (splitmix64, tests/oracle_lib.py).  Other formats: --format NAME (parity-test
            configs, reported to stderr / --formats-json, not the headline line).
  N > 1     the path shards by block rows with no data-path collective: every
            rank decodes its own 8192-row band of a 8192 x (8192*N) image -> "scaling": "weak".
            RCCL is used only for the timing barrier / max-reduction (and the optional gather,
            --gather, which is reported separately and is not part of `value`).
  value     Gpixel/s = N * width * height * K / max-over-ranks(wall time of K steps)
  roofline  algorithmic bytes per launch (blocks * (block_bytes + 16*pixel_bytes)) / average
            launch duration from HIP events recorded on the launch stream around the timed
            region; peak = 8 TB/s (MI355X_MICROARCH.md).
  cpu_baseline  the compiled reference (oracle/_ref, kind "reference") or our C restatement
            (kind "port") decoding the same stream on the host cores; rank 0, N == 1 only.
"""
import argparse
import ctypes
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(fmt, data, width, height, budget_s=12.0):
    """Time the CPU decode of the same block stream through detexDecompressTextureLinear
    (reference build if shipped, else the oracle port) on all host cores, row bands per thread
    (legal: all reference state is __thread).  Bounded to ~budget_s seconds."""
    import oracle_lib as ol
    cores = os.cpu_count() or 1
    wb, hb = width // 4, height // 4
    px, bs = fmt.pixel_bytes, fmt.block_bytes
    out = np.empty(width * height * px, np.uint8)
    if ol.have_ref():
        kind, api = "reference", ol.load_ref()

        def band(r0, r1):
            tex = ol.DetexTexture(fmt.texture_format, ol._ptr(data[r0 * wb * bs:]), width, (r1 - r0) * 4, wb, r1 - r0)
            api.lib.detexDecompressTextureLinear(ctypes.byref(tex), ol._ptr(out[r0 * 4 * width * px:]), fmt.texture_format & 0xFFFF)
    else:
        kind, orc = "port", ol.Oracle()

        def band(r0, r1):
            orc.lib.orc_decompress_linear(fmt.index, ol._ptr(data[r0 * wb * bs:]), width, (r1 - r0) * 4, wb, r1 - r0,
                                          ol._ptr(out[r0 * 4 * width * px:]))

    from concurrent.futures import ThreadPoolExecutor

    def run(pool, threads):
        # ctypes releases the GIL for the duration of each band's C call
        t0 = time.perf_counter()
        list(pool.map(lambda g: band(g * hb // threads, (g + 1) * hb // threads), range(threads)))
        return time.perf_counter() - t0

    t1 = min(run(ThreadPoolExecutor(1), 1) for _ in range(2))      # single thread
    best, best_threads, passes, spent = None, cores, 0, 0.0
    for threads in sorted({cores, max(1, cores // 2), max(1, cores // 4), min(cores, 32)}, reverse=True):
        with ThreadPoolExecutor(threads) as pool:
            run(pool, threads)                                      # warm the pool
            t_end = time.perf_counter() + budget_s / 4
            while time.perf_counter() < t_end:
                t = run(pool, threads)
                passes += 1
                spent += t
                if best is None or t < best:
                    best, best_threads = t, threads
    gp = width * height / 1e9
    return {"value": round(gp / best, 4), "unit": "Gpixel/s", "cores": best_threads, "kind": kind,
            "sample": "full %dx%d %s stream U through detexDecompressTextureLinear, best of %d passes, row bands on a "
                      "%d-thread pool (best of 4 pool sizes on %d host threads); 1 thread: %.4f Gpixel/s"
                      % (width, height, fmt.name, passes, best_threads, cores, gp / t1),
            "value_1thread": round(gp / t1, 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--format", default="BC1")
    ap.add_argument("--size", type=int, default=8192, help="image width = per-rank height in pixels")
    ap.add_argument("--band-height", type=int, default=0, help="per-rank band height if different from --size")
    ap.add_argument("--strong-image", type=int, default=0,
                    help="strong scaling: ONE S x S image sharded by block rows over the ranks (BASELINE north_star: 32768); "
                         "each rank decodes its S x S/N band, value = S*S*steps/time, \"scaling\": \"strong\"")
    ap.add_argument("--variant", type=int, default=0, help="kernel variant (include/detexhip.h)")
    ap.add_argument("--stream", default="U", choices=["U", "M"])
    ap.add_argument("--layout", default="linear", choices=["linear", "tiled"],
                    help="linear = detexDecompressTextureLinear (headline); tiled = detexDecompressTextureTiled (block-major output)")
    ap.add_argument("--target", default=None, help="target pixel format for the in-kernel epilogues: BGRA8, BGRX8, RGB8, FLOAT_BGRX16 (default: native)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--gather", action="store_true", help="also time the optional whole-image all-gather (N>1)")
    ap.add_argument("--formats-json", default=None, help="also bench every format, write a table to this path")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from detex_amd import binding, formats as F
    import oracle_lib as ol
    import streams

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world == 1 and args.gpus > 1:
        log("bench.py: --gpus %d needs `python -m torch.distributed.run --nproc-per-node %d`" % (args.gpus, args.gpus))
        sys.exit(2)
    if not torch.cuda.is_available():
        log("bench.py: no HIP device; the decode path has no CPU fallback")
        sys.exit(3)
    # DETEX_BENCH_BACKEND=gloo lets the N>1 code path be exercised on a 1-GPU box (all ranks share
    # cuda:0; a plumbing test, not a measurement).  The driver's runs use RCCL ("nccl").
    backend = os.environ.get("DETEX_BENCH_BACKEND", "nccl")
    device_index = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(device_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    binding.load()
    binding.set_kernel_variant(args.variant)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def make_input(fmt, W, H, seed_shift):
        data = ol.stream_u(fmt, (W // 4) * (H // 4), seed=ol.STREAM_SEED_BASE + ol.STREAM_SEED_K.get(fmt.name, 16 + fmt.index) + (seed_shift << 8))
        if args.stream == "M":
            data = streams.stream_m(fmt, data)
        return data

    TARGETS = {"BGRA8": F.PIXEL_FORMAT_BGRA8, "BGRX8": F.PIXEL_FORMAT_BGRX8, "RGB8": F.PIXEL_FORMAT_RGB8,
               "FLOAT_BGRX16": F.PIXEL_FORMAT_FLOAT_BGRX16, "RGBA8": F.PIXEL_FORMAT_RGBA8}

    def target_of(fmt):
        pf = TARGETS[args.target] if args.target else F.native_pixel_format(fmt)
        return pf, 1 + ((pf & 0xF00) >> 8)

    def run_format(fmt, W, H, steps, warmup):
        data = make_input(fmt, W, H, rank)
        d_blocks = torch.from_numpy(np.ascontiguousarray(data)).cuda()
        pf, tpx = target_of(fmt)
        d_out = torch.empty(W * H * tpx, dtype=torch.uint8, device="cuda")
        status = torch.zeros(1, dtype=torch.int32, device="cuda")
        if args.layout == "tiled":
            step = lambda: binding.decompress_tiled_device(fmt, d_blocks, W // 4, H // 4, out=d_out, status=status, pixel_format=pf)
        else:
            step = lambda: binding.decompress_linear_device(fmt, d_blocks, W, H, out=d_out, status=status, pixel_format=pf)
        for _ in range(warmup):
            step()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        t0 = time.perf_counter()
        e0.record()
        for _ in range(steps):
            step()
        e1.record()
        barrier()
        wall = time.perf_counter() - t0
        ev_ms = e0.elapsed_time(e1)
        if world > 1:
            t = torch.tensor([wall, ev_ms], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall, ev_ms = t.tolist()
        return data, d_blocks, d_out, wall, ev_ms / steps

    fmt = F.BY_NAME[args.format]
    W = H = args.size
    if args.band_height:
        H = args.band_height          # e.g. --size 32768 --band-height 4096: one GPU's band of a 32768^2 image over 8 GPUs
    if args.strong_image:
        from detex_amd import sharding
        shard = sharding.shard_of(rank, world, fmt, args.strong_image, args.strong_image)
        W, H = args.strong_image, (shard.row1 - shard.row0) * 4
    data, d_blocks, d_out, wall, launch_ms = run_format(fmt, W, H, args.steps, args.warmup)
    blocks = (W // 4) * (H // 4)
    pf, tpx = target_of(fmt)
    alg_bytes = blocks * (fmt.block_bytes + 16 * tpx)
    achieved = alg_bytes / (launch_ms * 1e-3) / 1e9
    gpix = world * W * H * args.steps / wall / 1e9

    gather = None
    if args.gather and world > 1:
        full = torch.empty(world * d_out.numel(), dtype=torch.uint8, device="cuda")
        dist.all_gather_into_tensor(full, d_out)
        barrier()
        t0 = time.perf_counter()
        for _ in range(5):
            dist.all_gather_into_tensor(full, d_out)
        barrier()
        gather = {"op": "all_gather_into_tensor", "ms": (time.perf_counter() - t0) / 5 * 1e3, "bytes_per_rank": d_out.numel()}

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    result = {
        "metric": "Gpixel/s decoded (%s -> %s, %dx%d per GPU, device-resident)" % (fmt.name, F.target_name(fmt), W, H),
        "value": round(gpix, 3), "unit": "Gpixel/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(wall / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "strong" if args.strong_image else "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "%s->%s %dx%d block stream %s (splitmix64 seed 0xD37E5000+k), one launch per step, "
                               "sharded by block rows: one %d-row band per GPU" % (fmt.name, F.target_name(fmt), W, H, args.stream, H),
                   "format": fmt.name, "width": W, "height_per_gpu": H, "blocks_per_gpu": blocks,
                   "kernel": binding.kernel_name(fmt) if args.layout == "linear" else "decode_blocks", "layout": args.layout, "variant": args.variant, "target_pixel_format": "0x%04X" % pf},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
                     "algorithmic_bytes_per_launch": alg_bytes, "launch_us": round(launch_ms * 1e3, 3),
                     "write_frac": round(blocks * 16 * tpx / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)},
    }
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            key = "%s/%d/v%d" % (fmt.name, W, args.variant) + ("/%s" % args.target if args.target else "")
            t = json.load(open(pmc)).get(key)
            if t:
                result["roofline"]["traffic"] = t["hbm_bytes_per_launch"]
                result["roofline"]["traffic_source"] = t.get("source")
        except Exception as e:  # noqa
            log("pmc_traffic.json unreadable:", e)
    if gather:
        result["gather"] = gather

    if world == 1:
        # bit-exactness of what was just timed, against the CPU checker on a bounded sample (first 64 block rows)
        rows = 64
        orc = ol.Oracle()
        if args.layout == "tiled":
            ok_o, want = orc.tiled_to(fmt, data[:rows * (W // 4) * fmt.block_bytes], W // 4, rows, pf)
        else:
            ok_o, want = orc.linear_to(fmt, data[:rows * (W // 4) * fmt.block_bytes], W, rows * 4, pf)
        got = d_out[:want.size].cpu().numpy()
        result["verified_bit_exact_rows"] = rows * 4 if np.array_equal(got, want) else 0
        if not np.array_equal(got, want):
            log("bench.py: OUTPUT MISMATCH against the oracle")
            result["value"] = 0.0
        # host-pointer drop-in tier (PCIe-inclusive; never `value`)
        try:
            if args.layout != "linear":
                raise RuntimeError("host tier is timed for the linear layout only")
            api = ol.DetexAPI(binding.LIB_PATH)
            host_out = np.empty(W * H * tpx, np.uint8)
            api.linear(fmt, data, W, H, out=host_out, pixel_format=pf)
            t0 = time.perf_counter(); api.linear(fmt, data, W, H, out=host_out, pixel_format=pf); th = time.perf_counter() - t0
            result["host_tier"] = {"gpixel_s_pcie_inclusive": round(W * H / th / 1e9, 3), "ms": round(th * 1e3, 2)}
        except Exception as e:  # noqa
            log("host tier timing failed:", e)
        if not args.no_cpu:
            result["cpu_baseline"] = cpu_baseline(fmt, data, W, H)

    if args.formats_json and world == 1:
        table = {}
        for f in F.FORMATS:
            for kind in (["U", "M"] if f.name in ("BPTC", "BPTC_FLOAT", "BPTC_SIGNED_FLOAT") else ["U"]):
                args.stream = kind
                args.target = None
                _, _, _, w_, ms_ = run_format(f, W, H, max(200, args.steps), 400)    # steady state: the first ~300 launches of a VALU-heavy kernel ride a power-management transient
                ab = blocks * (f.block_bytes + 16 * f.pixel_bytes)
                table["%s/%s" % (f.name, kind)] = {"launch_us": round(ms_ * 1e3, 2), "gpixel_s": round(W * H / (ms_ * 1e-3) / 1e9, 1),
                                                    "achieved_GBps": round(ab / (ms_ * 1e-3) / 1e9, 1),
                                                    "frac_of_8TBps": round(ab / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
                log(f.name, kind, table["%s/%s" % (f.name, kind)])
                torch.cuda.empty_cache()
        os.makedirs(os.path.dirname(os.path.abspath(args.formats_json)), exist_ok=True)
        json.dump(table, open(args.formats_json, "w"), indent=1, sort_keys=True)

    print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
