#!/usr/bin/env python3
"""bench.py -- throughput of the block-decode hot path on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W   (N>1 via torch.distributed.run)
prints ONE JSON line on rank 0.

  step      one pass of the hot path over one batch: one detexhipDecompressTextureLinearDevice
            call (= one kernel launch) decoding a whole block stream that is already resident in
            HBM into a device-resident linear image.  Before the W warm-up and K timed steps the
            same launch is repeated, untimed, until its duration has settled (`settle` in the
            line; --no-settle starts cold): the first few hundred launches after idle run through
            a power-management excursion that a steady decode stream never sees.
  N == 1    workload = BASELINE.json configs[1]: BC1 -> RGBA8, 8192 x 8192, synthetic stream U
            (splitmix64, tests/oracle_lib.py).  Extra keys: `per_format` (the six headline formats
            of configs[1..4] at 8192^2, streams U/M/C, plus the weakest kernels -- signed BC6H,
            block-major BC7, RGTC1 -- each timed at steady state), `beyond_mall`,
            `per_format.beyond_cache_16384` (the six headline formats, and the five formats whose 8192^2 footprint fits the Infinity Cache, at 16384^2), `cold` (first launch after idle, mean of the first
            twenty, the contract's W + K started cold), `roofline.blocks_from_hbm` / `per_format.formats.*.blocks_from_hbm` (the same launch
            over R different inputs in turn: the timed loop re-reads its ONE input's blocks from the 256 MiB Infinity Cache, this is every byte
            through HBM), `strong_image_32768` / `bc6h_32768_whole` (this GPU alone on the N>1 workloads: the WHOLE 32768^2 image through one call of
            the device entry, with the one-launch figure and a quarter-image band beside it), `host_tier`,
            `host_tier_small` (per-call latency of small textures / one block through the host API, beside the reference on one host thread;
            `oneshot_*`: a fresh process decoding the 17 bundled fixtures once), `cpu_baseline`.
  N > 1     BASELINE north_star: ONE 32768 x 32768 BC1 image sharded by block rows over the ranks
            (contiguous input and output ranges per rank, NO data-path collective)
            -> "scaling": "strong", value = 32768^2 * K / max-over-ranks(wall time of K steps).
            RCCL is used for the timing barrier / max-reduction only.  Extra keys: `weak` (one
            8192^2 image per rank), `gather` (the optional whole-image gather over xGMI, timed
            separately and never part of `value`: `to_root` = grouped point-to-point sends into one
            rank's image, `to_all` = one all_gather_into_tensor), `bc6h_32768` (BASELINE configs[4]:
            BPTC_FLOAT -> FLOAT_RGBX16, 32768^2 over the ranks, decode-only + its gathers, rank 0's
            EVERY rank's whole band checked against the reference's digests, as is every rank's band of the headline image:
            `whole_band_digests_match_reference_all_ranks`), `rccl_ranks` (ranks that answered an all_reduce; the run exits with
            code 5 unless that is N, one rank per distinct GPU).  --weak restores the round-1 line.  DETEX_BENCH_FORCE_DIST=1 takes
            this whole branch at ANY world size: with WORLD_SIZE 1 it is the RCCL pre-flight a one-GPU box can run (nccl process
            group, rank census, barriers, whole-image digests, both gathers, BC6H 32768^2): tests/test_gpu_rccl_preflight.py.
  roofline  algorithmic bytes per launch (blocks * (block_bytes + 16*pixel_bytes)) / average
            launch duration from HIP events recorded on the launch stream around the timed
            region; peak = 8 TB/s (MI355X_MICROARCH.md); `traffic` = HBM bytes per launch from
            rocprofv3 PMC counters collected DURING this run (live_pmc_traffic; `traffic_source` says "REPLAYED" when it had to fall
            back to profiles/pmc_traffic.json).  Measured in the same process beside it
            (tools/ubench/hbm_ref.hip): ref_fill_GBps = a write-only fill of 1 GiB with the decode
            kernels' store shape, ref_fill_same_shape_GBps = that fill over the workload's own
            output image, ref_copy_GBps = a 1 GiB 16-byte-vector copy (read + written);
            ref_fill_torch_GBps = torch's fill kernel over 1 GiB;
            frac_of_measured_fill = the kernel's WRITE rate / the best of those fills,
            frac_of_measured_copy = its read + write rate / ref_copy.  `beyond_mall`: the same
            format at 16384^2 (1 GiB of pixels: the 8192^2 output, 256 MiB, is exactly the size of
            the Infinity Cache).
  cpu_baseline  the compiled reference (oracle/_ref, kind "reference") or our C restatement
            (kind "port") decoding the same stream on the host cores; rank 0, N == 1 only.

Layout of this file: cpu_baseline / Telemetry / live_pmc_traffic / roofline_row (no GPU needed: tests/test_bench_helpers.py), Job and
RotatingInputs (device-resident input and output of one decode call), class Bench -- __init__: process group and workload geometry; helpers
(barrier, timed, steady_state_us, ...); one method per part of the line (measure_headline, multi_gpu_extras, headline_result, add_*); run() --
and main(): arguments, stdout hand-over, late imports.
"""
import argparse
import ctypes
import json
import os
import sys
import time

# (multi-process GPU work on this pool needs dmabuf IPC: the driver's environment exports this already; kept here, before the HIP runtime starts,
# for a launcher that hands the ranks a scrubbed environment -- without it RCCL fails with `hipIpcGetMemHandle: invalid argument`)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0
HEADLINE_FORMATS = ["BC1", "BC3", "BPTC", "ETC2", "ETC2_EAC", "BPTC_FLOAT"]
# formats whose 8192^2 footprint (blocks + pixels) fits the 256 MiB Infinity Cache: their HBM rate is measured at 16384^2 (384-768 MiB)
NARROW_FORMATS = ["RGTC1", "RGTC2", "SIGNED_RGTC1", "EAC_R11", "EAC_SIGNED_R11"]
# the kernels furthest below the roofline / with the shortest launches, reported beside the headline formats: (format, stream, layout)
# (stream F: signed BC6H has no fixture in the reference; the unsigned format's fixture, whose blocks are valid signed blocks too, stands in for
# coherent encoder-made content)
WEAK_KERNELS = [("BPTC_SIGNED_FLOAT", "U", "linear"), ("BPTC_SIGNED_FLOAT", "F", "linear"), ("BPTC", "U", "tiled"), ("RGTC1", "U", "linear")]


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


class Telemetry:
    """shader clock and package power of the GPU this process decodes on, read from the amdgpu driver's hwmon files (freq1_input in Hz,
    power1_input in microwatts: two small reads, ~20 us) by a sampling thread while a kernel loop runs; None where the files are absent."""

    def __init__(self, torch, device_index, hwmon_dir=None):
        self.dir = hwmon_dir
        if hwmon_dir is not None:
            return
        try:
            import glob
            p = torch.cuda.get_device_properties(device_index)
            bdf = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
            cand = glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*" % bdf)
            if not cand:                                     # (no PCI ids from torch: the one card rocm-smi would show)
                cand = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))[:1]
            for d in cand:
                if os.path.exists(os.path.join(d, "freq1_input")):
                    self.dir = d
                    break
        except Exception as e:  # noqa
            log("telemetry unavailable:", e)

    def _read(self, name):
        try:
            with open(os.path.join(self.dir, name)) as f:
                return int(f.read().strip())
        except Exception:  # noqa
            return None

    def during(self, fn, seconds=0.15):
        """run fn() repeatedly for `seconds` while sampling; returns {"sclk_mhz", "power_w", "samples"} of the second half of the samples"""
        if self.dir is None:
            return None
        import threading
        samples, stop = [], threading.Event()

        def poll():
            while not stop.is_set():
                samples.append((self._read("freq1_input"), self._read("power1_input")))
                time.sleep(0.002)
        t = threading.Thread(target=poll)
        t.start()
        t_end = time.perf_counter() + seconds
        while time.perf_counter() < t_end:
            fn()
        stop.set()
        t.join()
        tail = [x for x in samples[len(samples) // 2:] if x[0] is not None]
        if not tail:
            return None
        row = {"sclk_mhz": round(sum(x[0] for x in tail) / len(tail) / 1e6), "samples": len(tail)}
        pw = [x[1] for x in tail if x[1] is not None]
        if pw:
            row["power_w"] = round(sum(pw) / len(pw) / 1e6)
        return row


def live_pmc_traffic(fmt_name, side, layout="linear", timeout_s=150):
    """HBM bytes per launch of the decode kernel from rocprofv3 PMC counters, collected NOW on this box: two separate passes (FETCH_SIZE,
    WRITE_SIZE; --pmc with --kernel-trace only, as MI355X_MICROARCH.md's HBM section prescribes) over a child process that launches the same
    kernel a few times (tools/gpu_run_case.py); FETCH_SIZE is doubled (gfx950 tallies the 128-byte requests of a wide streaming read at 64 B).
    Returns a dict or None (tool absent, bench.py itself under a profiler, or a pass failed)."""
    import csv, glob, shutil, subprocess, tempfile
    if shutil.which("rocprofv3") is None:
        return None
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) for k in os.environ):     # bench.py is being profiled itself: no nested profiler
        return None
    kernel = "decode_linear" if layout == "linear" else "decode_blocks"
    out, tmp = {}, tempfile.mkdtemp(prefix="detex_pmc_", dir="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            cmd = ["rocprofv3", "--pmc", ctr, "--kernel-trace", "-d", d, "-o", "p", "--output-format", "csv", "--", sys.executable,
                   os.path.join(ROOT, "tools", "gpu_run_case.py"), fmt_name, "U", str(side), str(side), "0", "6", layout]
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
            rows = [row for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True) for row in csv.DictReader(open(f))
                    if kernel in row["Kernel_Name"] and row["Counter_Name"] == ctr]
            v = sorted(float(row["Counter_Value"]) for row in rows)
            if r.returncode != 0 or not v:
                return None
            out[ctr] = v[len(v) // 2]
            out["launches_" + ctr] = len(v)
        fetch, write = out["FETCH_SIZE"] * 1024 * 2, out["WRITE_SIZE"] * 1024
        return {"hbm_bytes_per_launch": int(fetch + write), "fetch_bytes": int(fetch), "write_bytes": int(write), "profiled_launches": out["launches_WRITE_SIZE"],
                "source": "measured in this run: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (two separate passes with --kernel-trace only) over tools/gpu_run_case.py, "
                          "median of the profiled %s launches; FETCH_SIZE KiB x 2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE KiB" % kernel}
    except Exception as e:  # noqa
        log("live PMC pass failed:", repr(e))
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


MALL_BYTES = 256 << 20                 # Infinity Cache (MI355X_MICROARCH.md): a footprint that fits is served from it, not from HBM


def roofline_row(alg_bytes, write_bytes, pixels, launch_us):
    """One row of the per-format tables from a launch time.  `frac` = algorithmic bytes / time / 8 TB/s -- but null where that is not an HBM
    fraction: a footprint (blocks + pixels) that fits the 256 MiB Infinity Cache, or a rate above the peak.  `write_frac` = pixels written /
    time / 8 TB/s (north_star's "HBM-write roofline"), which no cache inflates at sizes whose pixels do not fit it."""
    ach = alg_bytes / (launch_us * 1e-6) / 1e9
    resident = bool(alg_bytes <= MALL_BYTES)
    row = {"launch_us": round(launch_us, 2), "gpixel_s": round(pixels / (launch_us * 1e-6) / 1e9, 1),
           "achieved_GBps": round(ach, 1), "frac": None if resident else round(ach / HBM_PEAK_GBPS, 4),
           "footprint_MiB": round(alg_bytes / 2 ** 20, 1), "cache_resident": resident}
    if resident:
        row["frac_note"] = ("blocks + pixels fit the 256 MiB Infinity Cache: the rate is not an HBM rate (it may exceed 8 TB/s) and no roofline fraction is given; "
                            "per_format.beyond_cache_16384 has this format at a size that does not fit")
    row["write_frac"] = None if resident else round(write_bytes / (launch_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4)
    if not resident and ach > HBM_PEAK_GBPS:
        # (coherent content -- stream C -- in the cheapest decoders at 16384^2: 8.0-8.25 TB/s at the L2's memory interface, where the PMC counters see
        # exactly the algorithmic bytes for stream C as for stream U: profiles/r05/stream_c_16384_counters.jsonl.  The 128-256 MiB of BLOCKS
        # can be served by the Infinity Cache behind that interface from launch to launch, and the pins' rate is 8.19 TB/s, not the guide's
        # round 8: a rate above the peak is not an HBM rate, so no fraction is claimed -- write_frac stands)
        row["frac"] = None
        row["frac_note"] = ("above the 8 TB/s peak: the blocks of this size (<= 256 MiB) can be served by the Infinity Cache behind the L2's memory interface; "
                            "no HBM fraction is given, write_frac is the cache-proof figure")
    return row


# torch, torch.distributed and the package are imported in main(), after stdout has been handed to stderr (gloo and RCCL print there)
torch = dist = binding = F = sharding = ol = streams = None


class Job:
    """device-resident input/output of one decode call"""
    def __init__(self, fmt, W, H, data, layout, pf, tpx):
        self.fmt, self.W, self.H, self.data, self.layout = fmt, W, H, data, layout
        self.pf, self.tpx = pf, tpx
        self.d_blocks = torch.from_numpy(np.ascontiguousarray(data)).cuda()
        self.d_out = torch.empty(W * H * self.tpx, dtype=torch.uint8, device="cuda")
        self.status = torch.zeros(1, dtype=torch.int32, device="cuda")
        self.blocks = (W // 4) * (H // 4)
        self.alg_bytes = self.blocks * (fmt.block_bytes + 16 * self.tpx)

    def step(self):
        if self.layout == "tiled":
            binding.decompress_tiled_device(self.fmt, self.d_blocks, self.W // 4, self.H // 4, out=self.d_out, status=self.status, pixel_format=self.pf)
        else:
            binding.decompress_linear_device(self.fmt, self.d_blocks, self.W, self.H, out=self.d_out, status=self.status, pixel_format=self.pf)

    def verify(self, rows=64):
        """bit-exactness of what was just timed against the CPU checker on a bounded sample (first `rows` block rows)"""
        rows = min(rows, self.H // 4)
        orc = ol.Oracle()
        sub = self.data[:rows * (self.W // 4) * self.fmt.block_bytes]
        if self.layout == "tiled":
            _, want = orc.tiled_to(self.fmt, sub, self.W // 4, rows, self.pf)
        else:
            _, want = orc.linear_to(self.fmt, sub, self.W, rows * 4, self.pf)
        got = self.d_out[:want.size].cpu().numpy()
        return rows * 4 if np.array_equal(got, want) else 0


class RotatingInputs:
    """the same decode over R DIFFERENT input buffers in turn (R x blocks > 2.5 x the 256 MiB Infinity Cache), one output image: every
    launch's blocks come out of HBM -- the regime of a stream of different textures.  (A loop over ONE input re-reads its blocks from
    that memory-side cache: HBM then sees the writes only, and `frac` counts bytes HBM never delivered -- DESIGN.md section 4.)"""
    def __init__(self, job):
        self.job = job
        n = int(job.d_blocks.numel())
        self.inputs = [job.d_blocks] + [torch.roll(job.d_blocks, 4096 * k) for k in range(1, min(32, max(3, -(-(640 << 20) // n))))]    # (at most 32: tiny plumbing images)
        self.k = 0
        self.blocks, self.alg_bytes, self.tpx, self.W, self.H = job.blocks, job.alg_bytes, job.tpx, job.W, job.H

    def step(self):
        j = self.job
        self.k = (self.k + 1) % len(self.inputs)
        binding.decompress_linear_device(j.fmt, self.inputs[self.k], j.W, j.H, out=j.d_out, status=j.status, pixel_format=j.pf)


class Bench:
    """One run: the process-group set-up and the workload geometry (__init__), the measurement helpers, and one method per part of the JSON line
    (run() at the end calls them in order)."""

    def __init__(self, args):
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
        # (a launcher that shows every rank only its own GPU leaves one visible device per process: device 0 is then the rank's)
        ndev = torch.cuda.device_count()
        device_index = local_rank if (backend == "nccl" and local_rank < ndev) else local_rank % ndev
        torch.cuda.set_device(device_index)
        rccl_ranks = 1
        # DETEX_BENCH_FORCE_DIST=1: run the N > 1 code path (process group, rank census, barriers, band digests, both gathers) at ANY world size
        # -- with WORLD_SIZE 1 and the nccl backend this is the pre-flight of the RCCL path on a one-GPU box (tests/test_gpu_rccl_preflight.py)
        multi = world > 1 or os.environ.get("DETEX_BENCH_FORCE_DIST") == "1"
        if multi:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            # The census comes FIRST and travels over gloo (CPU tensors of a mixed-backend group): which GPU each rank sits on, by PCI address.
            # Two ranks on one GPU -- torchrun --nproc-per-node N on a box with fewer GPUs -- would make RCCL abort inside its communicator
            # set-up ("Duplicate GPU detected"); found here, before any RCCL call, the run ends with a message and exit code 5 instead.
            dist.init_process_group("cpu:gloo,cuda:nccl" if backend == "nccl" else backend, rank=rank, world_size=world)
            prop = torch.cuda.get_device_properties(device_index)
            mine = torch.tensor([device_index, getattr(prop, "pci_bus_id", -1), getattr(prop, "pci_device_id", -1), getattr(prop, "pci_domain_id", -1)], dtype=torch.int64)
            seen = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(seen, mine)
            places = [tuple(int(v) for v in t.tolist()) for t in seen]
            # distinct GPUs: by PCI address where the runtime reports one, else by device index (which then must not have been folded)
            by_pci = all(p[1] >= 0 for p in places)
            distinct = len({p[1:] for p in places}) == world if by_pci else (len({p[0] for p in places}) == world and world <= ndev)
            if backend == "nccl" and (args.gpus != world or not distinct):
                log("bench.py: rank %d: NOT one rank per GPU: --gpus %d, WORLD_SIZE %d, visible devices %d, (device, pci bus, pci device, pci domain) per rank %s"
                    % (rank, args.gpus, world, torch.cuda.device_count(), places))
                dist.destroy_process_group()
                sys.exit(5)
            ones = torch.ones(1, dtype=torch.int32, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(ones)                  # (the first CUDA collective: RCCL builds its communicator here)
            rccl_ranks = int(ones.item())          # ranks the collective library actually reached
            if backend == "nccl" and rccl_ranks != world:
                log("bench.py: rank %d: the all_reduce reached %d of %d ranks" % (rank, rccl_ranks, world))
                dist.destroy_process_group()
                sys.exit(5)
        binding.load()
        binding.set_kernel_variant(args.variant)
        coll_dev = "cuda" if backend == "nccl" else "cpu"
        telemetry = Telemetry(torch, device_index)
        fmt = F.BY_NAME[args.format]
        # (plumbing runs over gloo move CUDA tensors through the host at ~0.03 GB/s point-to-point: they get small images)
        big = 32768 if backend == "nccl" else 2048
        strong = args.strong_image if args.strong_image is not None else (0 if (not multi or args.weak) else big)
        W = H = args.size
        if args.band_height:
            H = args.band_height          # e.g. --size 32768 --band-height 4096: one GPU's band of a 32768^2 image over 8 GPUs
        if strong:
            shard = sharding.shard_of(rank, world, fmt, strong, strong)
            W, H = strong, (shard.row1 - shard.row0) * 4
        if not strong:
            shard = None
        self.args, self.world, self.rank, self.local_rank, self.backend, self.device_index, self.multi, self.rccl_ranks = args, world, rank, local_rank, backend, device_index, multi, rccl_ranks
        self.coll_dev, self.telemetry, self.fmt, self.big, self.strong, self.W, self.H, self.shard = coll_dev, telemetry, fmt, big, strong, W, H, shard
        self.extras = {}

    # ---- helpers ------------------------------------------------------------------------------------------------------------------------------
    def barrier(self):
        backend, device_index, multi = self.backend, self.device_index, self.multi
        torch.cuda.synchronize()
        if multi:
            if backend == "nccl":
                dist.barrier(device_ids=[device_index])      # (over RCCL, on this rank's GPU -- not the gloo half of the mixed-backend group)
            else:
                dist.barrier()
            torch.cuda.synchronize()

    def target_of(self, fmt, target=None):
        TARGETS = {"BGRA8": F.PIXEL_FORMAT_BGRA8, "BGRX8": F.PIXEL_FORMAT_BGRX8, "RGB8": F.PIXEL_FORMAT_RGB8,
                   "FLOAT_BGRX16": F.PIXEL_FORMAT_FLOAT_BGRX16, "RGBA8": F.PIXEL_FORMAT_RGBA8}
        pf = TARGETS[target] if target else F.native_pixel_format(fmt)
        return pf, 1 + ((pf & 0xF00) >> 8)

    def stream_seed(self, fmt, shift):
        return ol.STREAM_SEED_BASE + ol.STREAM_SEED_K.get(fmt.name, 16 + fmt.index) + (shift << 8)

    def make_input(self, fmt, wb, hb, kind, seed_shift=0):
        return streams.make_stream(kind, fmt, wb, hb, seed=self.stream_seed(fmt, seed_shift))

    def new_job(self, fmt, W, H, data, layout="linear", target=None):
        pf, tpx = self.target_of(fmt, target)
        return Job(fmt, W, H, data, layout, pf, tpx)

    def fresh_blocks_us(self, rot, producer, steps=80):
        """median launch time (HIP events around each launch alone) when the launch's blocks were WRITTEN right before it, on the same stream,
        into the rotating buffer it reads: `upload` = a copy out of pinned host memory (a texture streamer's upload -> decode), `device_copy` =
        a device-to-device copy (blocks produced on the GPU)"""
        j = rot.job
        src = j.d_blocks.cpu().pin_memory() if producer == "upload" else j.d_blocks.clone()

        def once(k, events):
            dst = rot.inputs[1 + k % (len(rot.inputs) - 1)]          # (inputs[0] is the job's own buffer: left alone)
            dst.copy_(src, non_blocking=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            binding.decompress_linear_device(j.fmt, dst, j.W, j.H, out=j.d_out, status=j.status, pixel_format=j.pf)
            e1.record()
            if events is not None:
                events.append((e0, e1))
        for k in range(20):
            once(k, None)
        torch.cuda.synchronize()
        ev = []
        for k in range(steps):
            once(k, ev)
        torch.cuda.synchronize()
        us = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
        return us[len(us) // 2]

    def blocks_from_hbm_row(self, job, producers=False):
        """launch time with the blocks coming out of HBM (rotating inputs), as shipped and with the read-ahead pass forced (detexhipSetReadAhead(2));
        `producers`: also with the blocks written right before each launch by an upload / by a device copy (which of the two regimes a pipeline sees)"""
        rot = RotatingInputs(job)
        us, _ = self.steady_state_us(rot, window=60, max_windows=8, min_launches=240, min_ms=20.0)
        row = {"inputs": len(rot.inputs), "launch_us": round(us, 2), "frac": round(job.alg_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4)}
        binding.set_read_ahead(2)
        try:
            us2, _ = self.steady_state_us(rot, window=60, max_windows=8, min_launches=240, min_ms=20.0)
        finally:
            binding.set_read_ahead(1)
        row["read_ahead_forced_launch_us"] = round(us2, 2)
        row["read_ahead_forced_frac"] = round(job.alg_bytes / (us2 * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4)
        if producers:
            try:
                row["after_upload_launch_us"] = round(self.fresh_blocks_us(rot, "upload"), 2)
                row["after_device_copy_launch_us"] = round(self.fresh_blocks_us(rot, "device_copy"), 2)
                row["producers_note"] = ("median of 80 launches timed one by one, the launch's blocks written into its (rotating) input buffer right before it on the "
                                         "same stream: uploaded blocks arrive as cold as any (DMA writes do not stay in the Infinity Cache), blocks copied by a kernel "
                                         "are partly served from it (DESIGN.md section 4)")
            except Exception as e:           # an extra; never the line
                row["producers_note"] = "not measured: %s" % (str(e)[:200],)
        del rot
        torch.cuda.empty_cache()
        return row

    def timed(self, job, steps, warmup):
        """the contract's timed region: W warm-up launches, then exactly K launches between barriers; wall = max over ranks"""
        multi, coll_dev = self.multi, self.coll_dev
        for _ in range(warmup):
            job.step()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.barrier()
        t0 = time.perf_counter()
        e0.record()
        for _ in range(steps):
            job.step()
        e1.record()
        self.barrier()
        wall = time.perf_counter() - t0
        ev_ms = e0.elapsed_time(e1)
        if multi:
            t = torch.tensor([wall, ev_ms], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall, ev_ms = t.tolist()
        return wall, ev_ms / steps

    def steady_state_us(self, job, window=100, max_windows=16, tol=0.012, min_launches=600, min_ms=40.0):
        """launch time once the power-management excursion has passed (profiles/AB_RECORD.md, DESIGN.md section 6: 20-40 % slower for roughly the 2nd to
        12th millisecond after idle -- VALU-heavy kernels and, since round 3, every kernel with `sc1 nt` stores -- then a slow approach
        to the settled clock): windows of launches (>= `window` of them and >= 4 ms each) until two consecutive ones agree within
        `tol` and at least `min_launches` launches and `min_ms` of kernel time have gone by; independent of the driver's --steps /
        --warmup.  (Round 2's first tables used 3 % / 400 launches and read VALU-heavy kernels up to 8 % high; a launch-count-only
        criterion let the 13 us RGTC1 kernel stop after 8 ms, in the middle of the excursion.)"""
        prev, done, us, spent_ms = None, 0, None, 0.0
        for _ in range(4 * max_windows):
            n = window if us is None else max(window, int(4000.0 / us) + 1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                job.step()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            us = ms / n * 1e3
            done += n
            spent_ms += ms
            if prev is not None and done >= min_launches and spent_ms >= min_ms and abs(us - prev) <= tol * prev:
                break
            if spent_ms > 400.0:
                break
            prev = us
        return us, done

    def roofline_of(self, job, launch_us, clocks=True):
        telemetry = self.telemetry
        row = roofline_row(job.alg_bytes, job.blocks * 16 * job.tpx, job.W * job.H, launch_us)
        if clocks:                         # shader clock and board power while this kernel runs back to back (0.15 s, hwmon files)
            t = telemetry.during(job.step)
            torch.cuda.synchronize()
            if t:
                row.update(t)
        return row

    def pmc_traffic(self, key):
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        try:
            return json.load(open(pmc)).get(key)
        except Exception:  # noqa
            return None

    def hbm_reference(self, job):
        """write-only fill and copy rates of this box, this process (reference points for the roofline fraction)"""
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import hbmref
            ref = {}
            # (each fill with all workgroups resident and with five per CU -- the cap the decode launches use -- the better one counts)
            ref["ref_fill_GBps"] = round(max(hbmref.fill_image_GBps(65536, 16384, 2, 20, 0, wg)[0] for wg in (0, 5)), 1)
            row_bytes = job.W * job.tpx
            if job.layout == "linear" and row_bytes % 4096 == 0 and job.H % 4 == 0:
                ref["ref_fill_same_shape_GBps"] = round(max(hbmref.fill_image_GBps(row_bytes, job.H, 2, 40, 0, wg)[0] for wg in (0, 5)), 1)
            c, _ = hbmref.copy_GBps(1 << 30, True, 10)
            ref["ref_copy_GBps"] = round(c, 1)
            a = torch.empty(1 << 28, dtype=torch.int32, device="cuda")             # torch's own fill kernel (ordinary stores) over 1 GiB
            ref["ref_fill_torch_GBps"] = round((1 << 30) / (hbmref.time_us(lambda: a.fill_(7), 20) * 1e-6) / 1e9, 1)
            del a
            ref["ref_note"] = ("same process: 1 GiB image-layout fill (four non-temporal 16-byte stores per lane, 1 KiB runs), the same fill "
                               "over this workload's output image, 1 GiB copy (bytes read + written), torch.Tensor.fill_ over 1 GiB; "
                               "frac_of_measured_fill = the kernel's write rate / the best of the three fills")
            torch.cuda.empty_cache()
            return ref
        except Exception as e:  # noqa
            log("hbm reference legs failed:", e)
            return {}

    def band_stream(self, f, image_side, sh, kind="U"):
        """this rank's band of the image's block stream: splitmix64 is counter-based (word k depends on k only), so the
        band is the slice [row0 * words_per_row, row1 * words_per_row) and a rank materialises nothing else"""
        words_per_row = (image_side // 4) * f.block_bytes // 8
        with np.errstate(over="ignore"):
            k = np.arange(sh.row0 * words_per_row + 1, sh.row1 * words_per_row + 1, dtype=np.uint64)
            z = np.uint64(self.stream_seed(f, 0)) + np.uint64(0x9E3779B97F4A7C15) * k
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
        d = z.view(np.uint8)
        return streams.stream_m(f, d) if kind == "M" else d

    def band_digests_match(self, f, side, sh, d_out):
        """this rank's whole band against the compiled reference's digests, eighth by eighth (tests/golden/digests_8192.json bands_all:
        the 32768^2 images of BC1 and BPTC_FLOAT in eight bands; a band of world N | 8 ranks is 8 / N consecutive eighths).  None where
        no golden applies."""
        world = self.world
        import hashlib
        try:
            gold = json.load(open(os.path.join(ROOT, "tests", "golden", "digests_8192.json")))["bands_all"]
        except Exception:  # noqa
            return None
        if side != 32768 or world not in (1, 2, 4, 8) or ("%s/32768/0of8" % f.name) not in gold:
            return None
        per = 8 // world
        ok = True
        for e in range(sh.rank * per, (sh.rank + 1) * per):
            g = gold["%s/32768/%dof8" % (f.name, e)]
            piece = d_out[(e - sh.rank * per) * g["bytes"]:(e - sh.rank * per + 1) * g["bytes"]]
            ok = ok and hashlib.sha256(piece.cpu().numpy().tobytes()).hexdigest() == g["sha256"]
        return ok

    def time_gathers(self, f, side, sh, band, reps=2):
        """the optional whole-image gather, timed separately from the decode (never part of `value`): to ONE rank with grouped
        point-to-point sends (the root's links to all peers busy at once, nothing lands elsewhere) and to EVERY
        rank with one all_gather_into_tensor straight into the final image"""
        world, rank = self.world, self.rank
        out = {}
        try:
            ok, image = sharding.gather_image_to_root(dist, torch, f, side, side, sh, band, True)
            self.barrier(); t0 = time.perf_counter()
            for _ in range(reps):
                ok, image = sharding.gather_image_to_root(dist, torch, f, side, side, sh, band, True, image=image)
            self.barrier(); ms = (time.perf_counter() - t0) / reps * 1e3
            row = {"op": "sharding.gather_image_to_root (grouped isend / irecv into the root's image)", "ms": round(ms, 3), "bytes_per_rank": int(sh.out_bytes)}
            if rank == 0:
                other = sharding.shard_of(world - 1, world, f, side, side)
                row["GBps_into_root"] = round((image.numel() - sh.out_bytes) / (ms * 1e-3) / 1e9, 1)
                row["own_band_intact"] = bool(torch.equal(image[sh.out_offset:sh.out_offset + sh.out_bytes], band[:sh.out_bytes]))
                row["peer_band_nonzero"] = bool(image[other.out_offset:other.out_offset + 4096].any().item())
            out["to_root"] = row
            del image
        except Exception as e:  # noqa
            out["to_root"] = {"error": repr(e)}
        torch.cuda.empty_cache()
        try:
            ok, image = sharding.gather_image(dist, torch, f, side, side, sh, band, True)
            self.barrier(); t0 = time.perf_counter()
            for _ in range(reps):
                ok, image = sharding.gather_image(dist, torch, f, side, side, sh, band, True, image=image)
            self.barrier(); ms = (time.perf_counter() - t0) / reps * 1e3
            other = sharding.shard_of((rank + 1) % world, world, f, side, side)
            out["to_all"] = {"op": "sharding.gather_image (all_gather_into_tensor into the final image)", "ms": round(ms, 3), "image_bytes": int(image.numel()),
                             "GBps_received_per_rank": round((image.numel() - sh.out_bytes) / (ms * 1e-3) / 1e9, 1),
                             "own_band_intact": bool(torch.equal(image[sh.out_offset:sh.out_offset + sh.out_bytes], band[:sh.out_bytes])),
                             "peer_band_nonzero": bool(image[other.out_offset:other.out_offset + 4096].any().item())}
            del image
        except Exception as e:  # noqa
            out["to_all"] = {"error": repr(e)}
        torch.cuda.empty_cache()
        return out

    # ---- the parts of the line ----------------------------------------------------------------------------------------------------------------
    def measure_headline(self):
        """the contract's part: this rank's input and output, the cold figures, the settling launches, W + K timed launches, verification"""
        args, world, rank, multi, coll_dev, fmt, strong, W, H, shard, extras = self.args, self.world, self.rank, self.multi, self.coll_dev, self.fmt, self.strong, self.W, self.H, self.shard, self.extras
        if strong and args.stream != "C":
            data = self.band_stream(fmt, strong, shard, args.stream)
        else:
            data = self.make_input(fmt, W // 4, H // 4, args.stream, 0 if strong else rank)
            if data is None:
                log("bench.py: stream C needs a bundled fixture of", fmt.name)
                sys.exit(4)
        job = self.new_job(fmt, W, H, data, args.layout, args.target)
        # Before the contract's W + K launches: run the kernel until its launch time has settled (the same criterion as the per-format
        # table).  The first ~250 launches after idle are not representative of a decode stream -- VALU-heavy kernels slow down for
        # a few hundred launches while the power management reacts, and the `sc1 nt` row stores of round 3 show the same excursion
        # (BC1, 25-launch windows: 41.6 41.2 44.0 48.6 47.9 46.3 44.4 43.1 42.3 41.2 40.9 41.1 ...) -- so a
        # timed region of 20-200 launches right after start-up would measure the excursion, not the kernel.  --no-settle skips it.
        cold = {}
        if not args.no_settle:
            # what a caller who decodes ONE texture sees: the first launch after idle and the mean of the first twenty (an event per launch),
            # then -- idle again -- the contract's W + K launches started cold (`value_cold`: the measurement of rounds 1 and 2)
            job.step(); torch.cuda.synchronize()                                 # (the very first launch also loads the code object: not counted)
            time.sleep(0.3)
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
            evs[0].record()
            for k in range(20):
                job.step()
                evs[k + 1].record()
            torch.cuda.synchronize()
            per = [evs[k].elapsed_time(evs[k + 1]) * 1e3 for k in range(20)]
            cold["cold_first_launch_us"] = round(per[0], 2)
            cold["first_20_mean_us"] = round(sum(per) / 20, 2)
            time.sleep(0.3)
            wall_c, launch_ms_c = self.timed(job, args.steps, args.warmup)
            image_pixels_c = strong * strong if strong else world * W * H
            cold["value_cold"] = round(image_pixels_c * args.steps / wall_c / 1e9, 3)
            cold["ms_per_step_cold"] = round(wall_c / args.steps * 1e3, 5)
            cold["launch_us_cold"] = round(launch_ms_c * 1e3, 3)
            cold["note"] = ("started 0.3 s after the previous launch: the first launch, the mean of the first 20 (one HIP event pair each), and the contract's W + K "
                            "launches without the settling phase that precedes `value`")
        settle_us, settle_launches = (None, 0) if args.no_settle else self.steady_state_us(job)
        wall, launch_ms = self.timed(job, args.steps, args.warmup)
        image_pixels = strong * strong if strong else world * W * H
        gpix = image_pixels * args.steps / wall / 1e9
        achieved = job.alg_bytes / (launch_ms * 1e-3) / 1e9

        # what every rank just wrote, checked against the CPU oracle on a bounded sample; AND over ranks
        verified_rows = job.verify(16 if multi else 64)
        if multi:
            v = torch.tensor([verified_rows], dtype=torch.int32, device=coll_dev)
            dist.all_reduce(v, op=dist.ReduceOp.MIN)
            verified_rows = int(v.item())

        if strong == 32768 and args.stream == "U" and not args.target and args.layout == "linear":
            mine = self.band_digests_match(fmt, strong, shard, job.d_out)
            if mine is not None:
                flag = torch.tensor([1 if mine else 0], dtype=torch.int32, device=coll_dev)
                if multi:
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                extras["whole_band_digests_match_reference_all_ranks"] = bool(flag.item())
                if not mine:
                    log("bench.py: rank %d: WHOLE-BAND DIGEST MISMATCH against the compiled reference" % rank)
        self.data, self.job, self.cold, self.settle_us, self.settle_launches = data, job, cold, settle_us, settle_launches
        self.wall, self.launch_ms, self.gpix, self.achieved, self.verified_rows = wall, launch_ms, gpix, achieved, verified_rows

    def multi_gpu_extras(self):
        """N > 1 only: the optional gathers of the headline image, BASELINE configs[4] (BC6H 32768^2 over the ranks), the weak-scaling line"""
        args, world, rank, coll_dev, fmt, big, strong, shard, job, wall, extras = self.args, self.world, self.rank, self.coll_dev, self.fmt, self.big, self.strong, self.shard, self.job, self.wall, self.extras
        # (a) the optional whole-image gather of the headline image
        if strong:
            extras["gather"] = self.time_gathers(fmt, strong, shard, job.d_out)
            extras["gather"]["decode_ms_per_step"] = round(wall / args.steps * 1e3, 4)
        # (a'') the same band decode with its blocks coming out of HBM: from N = 2 on a band's blocks (<= 256 MiB) fit the Infinity Cache and the timed
        # loop above re-reads them from there, which the whole image on one GPU (512 MiB of blocks) cannot -- a strong-scaling curve taken from `value`
        # alone would be super-linear for that reason.  Here every rank decodes R different inputs of its band's shape in turn (same barriers, max over ranks).
        if strong:
            try:
                rot = RotatingInputs(job)
                r_wall, r_ms = self.timed(rot, args.steps, args.warmup)
                extras["blocks_from_hbm"] = {"value_gpixel_s": round(strong * strong * args.steps / r_wall / 1e9, 3), "ms_per_step": round(r_wall / args.steps * 1e3, 5),
                                             "launch_us": round(r_ms * 1e3, 3), "inputs_per_rank": len(rot.inputs),
                                             "note": "the headline's loop over R different inputs per rank in turn (every launch's blocks out of HBM): the figure to divide "
                                                     "by the N = 1 line's strong_image_32768 for a like-for-like strong-scaling curve (DESIGN.md sections 4 and 7)"}
                del rot
                job.step()                              # (the band's own pixels back in d_out)
                torch.cuda.synchronize()
            except Exception as e:  # noqa
                extras["blocks_from_hbm"] = {"error": repr(e)}
            torch.cuda.empty_cache()
        # (a') BASELINE configs[4]: BC6H -> FLOAT_RGBX16 ("FP16 RGBA"), 32768^2 sharded over the ranks, decode-only and gather separately
        if strong and fmt.name != "BPTC_FLOAT":
            try:
                f6 = F.BY_NAME["BPTC_FLOAT"]
                sh6 = sharding.shard_of(rank, world, f6, big, big)
                d6 = self.band_stream(f6, big, sh6)
                j6 = self.new_job(f6, big, (sh6.row1 - sh6.row0) * 4, d6)
                w6, ms6 = self.timed(j6, args.steps, max(args.warmup, 10))
                v6 = j6.verify(16)
                digest = self.band_digests_match(f6, big, sh6, j6.d_out)      # EVERY rank digests its whole band (eighths of the golden image)
                if digest is not None:
                    dflag = torch.tensor([1 if digest else 0], dtype=torch.int32, device=coll_dev)
                    dist.all_reduce(dflag, op=dist.ReduceOp.MIN)
                    digest = bool(dflag.item())
                v = torch.tensor([v6], dtype=torch.int32, device=coll_dev)
                dist.all_reduce(v, op=dist.ReduceOp.MIN)
                row = {"workload": "BPTC_FLOAT->FLOAT_RGBX16, ONE %dx%d image (stream U) sharded by block rows over %d GPU(s): %d rows per GPU, no data-path collective"
                                   % (big, big, world, (sh6.row1 - sh6.row0) * 4),
                       "value_gpixel_s": round(big * big * args.steps / w6 / 1e9, 3), "ms_per_step": round(w6 / args.steps * 1e3, 5), "launch_us": round(ms6 * 1e3, 3),
                       "frac": round(j6.alg_bytes / (ms6 * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "verified_bit_exact_rows_min_over_ranks": int(v.item()),
                       "whole_band_digests_match_reference_all_ranks": digest}
                row["gather"] = self.time_gathers(f6, big, sh6, j6.d_out, reps=1)
                extras["bc6h_32768"] = row
                del j6
            except Exception as e:  # noqa
                extras["bc6h_32768"] = {"error": repr(e)}
        torch.cuda.empty_cache()
        # (b) weak scaling: one 8192^2 image per rank
        if strong:
            wjob = self.new_job(fmt, args.size, args.size, self.make_input(fmt, args.size // 4, args.size // 4, args.stream if args.stream != "C" else "U", rank), args.layout, args.target)
            w_wall, w_ms = self.timed(wjob, args.steps, args.warmup)
            extras["weak"] = {"workload": "%s %dx%d per GPU" % (fmt.name, args.size, args.size), "value_gpixel_s": round(world * args.size * args.size * args.steps / w_wall / 1e9, 3),
                              "ms_per_step": round(w_wall / args.steps * 1e3, 5), "launch_us": round(w_ms * 1e3, 3)}
            del wjob

    def headline_result(self):
        """the contract's keys"""
        args, world, fmt, strong, W, H, job, cold, settle_us, settle_launches, wall, launch_ms, gpix, achieved, verified_rows, extras = self.args, self.world, self.fmt, self.strong, self.W, self.H, self.job, self.cold, self.settle_us, self.settle_launches, self.wall, self.launch_ms, self.gpix, self.achieved, self.verified_rows, self.extras
        tname = F.PIXEL_FORMAT_NAMES.get(job.pf, "0x%04X" % job.pf)
        if strong:
            workload = ("%s->%s, ONE %dx%d image (stream %s, splitmix64 seed 0xD37E5000+k) sharded by block rows over %d GPU(s): "
                        "one %d-row band per GPU, one launch per step, no data-path collective" % (fmt.name, tname, strong, strong, args.stream, world, H))
        else:
            workload = ("%s->%s %dx%d block stream %s (splitmix64 seed 0xD37E5000+k), one launch per step, one image per GPU"
                        % (fmt.name, tname, W, H, args.stream))
        result = {
            "metric": "Gpixel/s decoded (%s -> %s, %s, device-resident)" % (fmt.name, tname, ("%dx%d image over %d GPU(s)" % (strong, strong, world)) if strong else "%dx%d per GPU" % (W, H)),
            "value": round(gpix, 3), "unit": "Gpixel/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(wall / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": workload, "format": fmt.name, "width": W, "height_per_gpu": H, "blocks_per_gpu": job.blocks, "stream": args.stream,
                       "kernel": binding.kernel_name(fmt) if args.layout == "linear" else "decode_blocks", "layout": args.layout, "variant": args.variant,
                       "target_pixel_format": "0x%04X" % job.pf},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
                         "algorithmic_bytes_per_launch": job.alg_bytes, "launch_us": round(launch_ms * 1e3, 3),
                         "write_frac": round(job.blocks * 16 * job.tpx / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                         "note": "frac = algorithmic bytes (blocks read once + pixels written once) / launch time / 8 TB/s.  The timed loop decodes ONE input again and "
                                 "again, so the blocks' share of those bytes is re-read from the 256 MiB memory-side Infinity Cache, not from HBM: write_frac is the HBM "
                                 "part of this loop, blocks_from_hbm.frac the same launch with every byte through HBM (DESIGN.md section 4)"},
            "verified_bit_exact_rows": verified_rows,
            "settle": {"launches_before_warmup": settle_launches, "last_window_us": None if settle_us is None else round(settle_us, 3),
                       "note": "untimed launches before the W warm-up steps, until two 100-launch windows agree within 1.2 % and >= 600 ran (--no-settle: none)"},
        }
        if cold:
            result["cold"] = cold
        if verified_rows == 0 or extras.get("whole_band_digests_match_reference_all_ranks") is False:
            log("bench.py: OUTPUT MISMATCH against the oracle / the reference's band digests")
            result["value"] = 0.0
        return result

    def add_roofline_references(self, result):
        """N = 1: clock / power, the fill and copy references of this process, the same launch with its blocks coming out of HBM"""
        args, telemetry, fmt, job, launch_ms, achieved = self.args, self.telemetry, self.fmt, self.job, self.launch_ms, self.achieved
        t = telemetry.during(job.step)
        torch.cuda.synchronize()
        if t:
            result["roofline"].update(t)
        ref = self.hbm_reference(job)
        result["roofline"].update(ref)
        if args.layout == "linear":
            try:
                result["roofline"]["blocks_from_hbm"] = dict(self.blocks_from_hbm_row(job, producers=True), note=(
                    "the timed loop decodes ONE input again and again: its %d MiB of blocks are re-read from the 256 MiB memory-side Infinity Cache (which the L2's "
                    "EA counters behind `traffic` count as memory requests), HBM itself sees the pixel writes (`write_frac`).  Here the same launch over R different "
                    "inputs in turn, so that every launch's blocks come out of HBM; read_ahead_forced_*: with detexhipSetReadAhead(2) -- a read-only pass over the blocks, "
                    "then the decode" % (job.blocks * fmt.block_bytes >> 20)))
            except Exception as e:  # noqa
                log("blocks_from_hbm failed:", e)
        write_gbps = job.blocks * 16 * job.tpx / (launch_ms * 1e-3) / 1e9
        result["roofline"]["write_GBps"] = round(write_gbps, 1)
        best_fill = max(ref.get("ref_fill_same_shape_GBps", 0), ref.get("ref_fill_GBps", 0), ref.get("ref_fill_torch_GBps", 0))
        if best_fill:
            result["roofline"]["frac_of_measured_fill"] = round(write_gbps / best_fill, 4)
        if ref.get("ref_copy_GBps"):
            result["roofline"]["frac_of_measured_copy"] = round(achieved / ref["ref_copy_GBps"], 4)

    def add_traffic(self, result):
        """HBM bytes per launch from rocprofv3 counter passes made now (or replayed from profiles/)"""
        args, multi, fmt, W, H, job = self.args, self.multi, self.fmt, self.W, self.H, self.job
        live = None
        if not multi and not args.no_extras and not args.target and W == H and args.stream == "U":
            live = live_pmc_traffic(fmt.name, W, args.layout)
        if live:
            result["roofline"]["traffic"] = live["hbm_bytes_per_launch"]
            result["roofline"]["traffic_source"] = live["source"]
            result["roofline"]["traffic_detail"] = {k: live[k] for k in ("fetch_bytes", "write_bytes", "profiled_launches")}
            result["roofline"]["traffic_over_algorithmic"] = round(live["hbm_bytes_per_launch"] / job.alg_bytes, 4)
        else:
            t = self.pmc_traffic("%s/%d/%s" % (fmt.name, W, args.layout) + ("/%s" % args.target if args.target else ""))
            if t:
                result["roofline"]["traffic"] = t["hbm_bytes_per_launch"]
                result["roofline"]["traffic_source"] = "REPLAYED from profiles/pmc_traffic.json (no live counter pass in this run): " + str(t.get("source"))

    def add_collective_keys(self, result):
        world, backend, rccl_ranks = self.world, self.backend, self.rccl_ranks
        result["rccl_ranks"] = rccl_ranks
        result["forced_dist_path"] = bool(world == 1)            # DETEX_BENCH_FORCE_DIST=1: the N > 1 code path at world size 1 (pre-flight, not a scaling point)
        try:
            result["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None
        except Exception:  # noqa
            result["rccl_version"] = None
        result["collective_backend"] = backend

    def add_host_tier(self, result):
        """host-pointer drop-in tier (PCIe-inclusive; never `value`)"""
        args, fmt, W, H, data, job = self.args, self.fmt, self.W, self.H, self.data, self.job
        # host-pointer drop-in tier (PCIe-inclusive; never `value`)
        try:
            if args.layout != "linear":
                raise RuntimeError("host tier is timed for the linear layout only")
            api = ol.DetexAPI(binding.LIB_PATH)
            host_out = np.empty(W * H * job.tpx, np.uint8)
            api.linear(fmt, data, W, H, out=host_out, pixel_format=job.pf)
            best = None
            for _ in range(3):
                t0 = time.perf_counter(); api.linear(fmt, data, W, H, out=host_out, pixel_format=job.pf); th = time.perf_counter() - t0
                best = th if best is None or th < best else best
            result["host_tier"] = {"gpixel_s_pcie_inclusive": round(W * H / best / 1e9, 3), "ms": round(best * 1e3, 2)}
        except Exception as e:  # noqa
            log("host tier timing failed:", e)

    def add_per_format(self, result):
        """per-format tables: the headline formats at 8192^2 (streams U / M / C) and the weakest kernels; the same formats and the narrow ones at 16384^2"""
        args, job = self.args, self.job
        # per-format table of the headline formats (BASELINE configs[1..4]) at 8192^2, steady state, streams U / M / C
        table = {}
        t_start = time.perf_counter()
        for name in HEADLINE_FORMATS:
            f = F.BY_NAME[name]
            for kind in (["U", "M", "C"] if name in ("BPTC", "BPTC_FLOAT") else ["U", "C"]):
                d = self.make_input(f, 2048, 2048, kind)
                if d is None:
                    continue
                j = self.new_job(f, 8192, 8192, d)
                us, launches = self.steady_state_us(j)
                row = self.roofline_of(j, us)
                row["launches_before_reading"] = launches
                tr = self.pmc_traffic("%s/8192/linear" % name)
                if tr and kind == "U":
                    row["traffic"] = tr["hbm_bytes_per_launch"]
                if kind == "U":
                    try:
                        row["blocks_from_hbm"] = self.blocks_from_hbm_row(j)
                    except Exception as e:  # noqa
                        log("blocks_from_hbm failed:", name, e)
                table["%s/%s" % (name, kind)] = row
                del j
        for name, kind, layout in WEAK_KERNELS:
            f = F.BY_NAME[name]
            j = self.new_job(f, 8192, 8192, self.make_input(f, 2048, 2048, kind), layout)
            us, launches = self.steady_state_us(j)
            row = self.roofline_of(j, us)
            row["launches_before_reading"] = launches
            table["%s/%s" % (name, kind) + ("/tiled" if layout == "tiled" else "")] = row
            del j
        torch.cuda.empty_cache()
        if result["roofline"].get("ref_copy_GBps"):     # a mixed read + write stream: beside the 8 TB/s fraction, the fraction of the copy measured in this process
            for row in table.values():
                if row["frac"] is not None:
                    row["frac_of_measured_copy"] = round(row["achieved_GBps"] / result["roofline"]["ref_copy_GBps"], 4)
        # the six headline formats beyond the Infinity Cache: 16384^2 (1 GiB of 32-bit pixels, 2 GiB of BC6H's)
        big_table = {}
        for name in HEADLINE_FORMATS + NARROW_FORMATS:
            try:
                f = F.BY_NAME[name]
                j = self.new_job(f, 16384, 16384, self.make_input(f, 4096, 4096, "U"))
                us, launches = self.steady_state_us(j, window=25, max_windows=8, min_launches=100)
                row = self.roofline_of(j, us)
                row["launches_before_reading"] = launches
                tr = self.pmc_traffic("%s/16384/linear" % name)
                if tr:
                    row["traffic"] = tr["hbm_bytes_per_launch"]
                    row["traffic_over_algorithmic"] = round(tr["hbm_bytes_per_launch"] / j.alg_bytes, 4)
                big_table["%s/U" % name] = row
                del j
            except Exception as e:  # noqa
                big_table["%s/U" % name] = {"error": repr(e)}
            torch.cuda.empty_cache()
        if result["roofline"].get("ref_copy_GBps"):
            for row in big_table.values():
                if row.get("frac") is not None:
                    row["frac_of_measured_copy"] = round(row["achieved_GBps"] / result["roofline"]["ref_copy_GBps"], 4)
        result["per_format"] = {"size": "8192x8192", "note": "launch time at steady state (windows of 100 launches until two agree within 1.2 % and >= 600 launches ran); "
                                                             "sclk_mhz / power_w: hwmon samples while the kernel then runs back to back for 0.15 s; cache_resident: "
                                                             "blocks + pixels <= the 256 MiB Infinity Cache",
                                "seconds": round(time.perf_counter() - t_start, 2), "formats": table,
                                "beyond_cache_16384": {"size": "16384x16384", "formats": big_table}}
        # (the headline format's row of that table under its round-3 name)
        if "launch_us" in big_table.get("%s/U" % args.format, {}):
            row = dict(big_table["%s/U" % args.format])
            row["workload"] = "%s 16384x16384 stream U" % args.format
            best_fill = max(result["roofline"].get(k, 0) for k in ("ref_fill_GBps", "ref_fill_torch_GBps"))
            if best_fill:
                row["frac_of_measured_fill"] = round(16384 * 16384 * job.tpx / (row["launch_us"] * 1e-6) / 1e9 / best_fill, 4)
            result["beyond_mall"] = row

    def add_whole_images(self, result):
        # this GPU alone on the N > 1 workloads: the WHOLE 32768^2 image of BASELINE's strong-scaling configuration (BC1; configs[4]'s BC6H image
        # likewise) through ONE call of the device entry, so that the driver's N = 1 and N > 1 values divide without a footnote.  Beside it: the
        # same call with the library's read-ahead switched off (one launch: the blocks -- 512 MiB / 1 GiB, more than the 256 MiB Infinity
        # Cache holds -- come out of HBM in the middle of the write stream) and a quarter-image band decoded alone (its 128 MiB of blocks are
        # re-read from that cache on every repetition: the rate an N = 4 rank of this benchmark sees, NOT a quarter of the image's time).
        for key, name in (("strong_image_32768", "BC1"), ("bc6h_32768_whole", "BPTC_FLOAT")):
            try:
                f = F.BY_NAME[name]
                whole = sharding.shard_of(0, 1, f, 32768, 32768)
                j = self.new_job(f, 32768, 32768, self.band_stream(f, 32768, whole))
                us, _ = self.steady_state_us(j, window=8, max_windows=6, min_launches=24, min_ms=20.0)
                verified = j.verify(16)
                binding.set_read_ahead(False)
                try:
                    us_one, _ = self.steady_state_us(j, window=8, max_windows=6, min_launches=24, min_ms=20.0)
                finally:
                    binding.set_read_ahead(True)
                row = {"workload": "%s 32768x32768 stream U, the whole image on this GPU in ONE detexhipDecompressTextureLinearDevice call" % name,
                       "whole_image_launch_us": round(us, 1), "gpixel_s": round(32768 * 32768 / (us * 1e-6) / 1e9, 1),
                       "frac": round(j.alg_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4), "verified_bit_exact_rows": verified,
                       "how": "bands of <= 128 MiB of blocks, each read into the Infinity Cache by a read-only pass and then decoded (detexhipSetReadAhead, on by default)",
                       "one_launch_us": round(us_one, 1), "one_launch_frac": round(j.alg_bytes / (us_one * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4)}
                del j
                torch.cuda.empty_cache()
                jb = self.new_job(f, 32768, 8192, self.band_stream(f, 32768, sharding.shard_of(0, 4, f, 32768, 32768)))
                us_b, _ = self.steady_state_us(jb, window=20, max_windows=6, min_launches=60)
                row.update({"band_32768x8192_launch_us": round(us_b, 2), "band_frac": round(jb.alg_bytes / (us_b * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4),
                            "band_note": "a quarter of the image decoded ALONE, repeatedly: its blocks stay in the Infinity Cache between repetitions, HBM sees writes only; "
                                         "four such bands of ONE image do not (DESIGN.md section 4)"})
                del jb
                result[key] = row
            except Exception as e:  # noqa
                log("%s failed:" % key, e)
                result[key] = {"error": repr(e)}
            torch.cuda.empty_cache()

    def add_host_tier_small(self, result):
        # small inputs through the reference's own entry points (host pointers): where the PCIe-attached decoder loses to one
        # host thread.  Per call, including the ctypes call overhead on both sides (~2 us).
        try:
            def per_call_us(fn, budget_s=0.2, min_calls=20):
                fn(); fn()
                t0 = time.perf_counter(); n = 0
                while n < min_calls or time.perf_counter() - t0 < budget_s:
                    fn(); n += 1
                return (time.perf_counter() - t0) / n * 1e6
            api = ol.DetexAPI(binding.LIB_PATH)
            ref_api = ol.load_ref() if ol.have_ref() else None
            small = {"note": "us per call: detexDecompressTextureLinear(BC1 -> RGBA8, host pointers) and the one-block leaf function, timed from Python "
                             "through ctypes (`gpu`, `reference_1thread`: ~2-3 us of call overhead each) and from a compiled C client; one block and textures of up to "
                             "1024 blocks go to a resident kernel from the second call in a row on, textures up to 1.25 MiB of blocks + pixels are exchanged "
                             "through pinned host memory (one launch, completion polled), larger ones staged through device buffers"}
            f1 = F.BY_NAME["BC1"]
            blk = ol.stream_u(f1, 1, seed=5)
            o16 = np.zeros(64, np.uint8)
            for label, a in (("gpu", api), ("reference_1thread", ref_api)):
                if a is None:
                    continue
                fn = a.block_fn(f1)
                row = {"one_block_us": round(per_call_us(lambda: fn(ol._ptr(blk), 0xFFFFFFFF, 0, ol._ptr(o16))), 2)}
                for side in (64, 256, 512, 1024):
                    d = ol.stream_u(f1, (side // 4) ** 2, seed=side)
                    o = np.empty(side * side * 4, np.uint8)
                    row["%dx%d_us" % (side, side)] = round(per_call_us(lambda: a.linear(f1, d, side, side, out=o)), 1)
                small[label] = row
            # the same calls from a compiled C program (tests/c_client/detex_client --latency): no ctypes overhead (~3 us per call above)
            client = os.path.join(ROOT, "tests", "c_client", "detex_client")
            if os.path.exists(client):
                import subprocess

                def client_latency(path, env=None):
                    r = subprocess.run([path, "--latency"], capture_output=True, text=True, timeout=120, env=env)
                    cc = {}
                    for line in r.stdout.splitlines():
                        if line.startswith("latency ") and "=" in line:
                            k, v = line.split()[1].split("=")
                            cc[k] = float(v)
                    return cc
                cc = client_latency(client)
                if cc:
                    cc["note"] = ("median of 2000 back-to-back calls, compiled C, same library: from the second call on the requests are answered by the resident "
                                  "kernel (detexhipSetResidentIdleMicroseconds; 256x256 is beyond it: one launch per call, completion polled)")
                    small["gpu_compiled_c_client"] = cc
                def client_latency_owned(path):
                    r = subprocess.run([path, "--latency", "owned"], capture_output=True, text=True, timeout=120)
                    return {l.split()[1].split("=")[0]: float(l.split()[1].split("=")[1]) for l in r.stdout.splitlines() if l.startswith("latency ") and "=" in l}
                cc = client_latency_owned(client)
                if cc:
                    cc["note"] = ("the same calls with pixel buffers from detexhipAllocPixelBuffer (pinned, device-visible): linear textures with up to 8 MiB of "
                                  "pixels are written by the kernel straight into the caller's buffer, nothing is copied out")
                    small["gpu_compiled_c_client_owned_pixel_buffers"] = cc
                cc = client_latency(client, dict(os.environ, DETEXHIP_RESIDENT_US="0"))
                if cc:
                    cc["note"] = "the same with DETEXHIP_RESIDENT_US=0: one launch per call, completion by polling a word the kernel releases in pinned memory"
                    small["gpu_compiled_c_client_launch_per_call"] = cc
                if os.path.exists(client + "_reflib"):
                    cc = client_latency(client + "_reflib")
                    if cc:
                        cc["note"] = "the same program linked against the compiled reference (oracle/_ref), one host thread"
                        small["reference_compiled_c_client"] = cc
                # the migration path of a per-block client: n independent blocks through ONE detexhipDecompressBlocks call against the loop over the
                # leaf function, in this library and in the compiled reference (tests/c_client/detex_client --blocks)
                def client_blocks(path):
                    r = subprocess.run([path, "--blocks"], capture_output=True, text=True, timeout=180)
                    rows = {}
                    for line in r.stdout.splitlines():
                        if line.startswith("blocks format="):
                            f = dict(kv.split("=") for kv in line.split()[1:])
                            rows["%s/%s" % (f["format"], f["n"])] = {k: (None if float(f[k]) < 0 else round(float(f[k]), 2)) for k in ("loop_us", "batched_us")}
                    return rows
                bb = client_blocks(client)
                if bb:
                    small["batched_blocks_compiled_c"] = {"gpu": bb, "note": "us per n blocks (BC1, BPTC; n = 1, 1024, 1048576): loop_us = n calls of the leaf function "
                                                          "detexDecompressBlock<FMT>, batched_us = ONE detexhipDecompressBlocks call (include/detexhip.h); a loop of 2^20 "
                                                          "trips to the GPU is not timed"}
                    if os.path.exists(client + "_reflib"):
                        small["batched_blocks_compiled_c"]["reference_1thread"] = client_blocks(client + "_reflib")
                # what a ONE-SHOT client pays (the reference's own callers decode their files once and exit: validate.c:188-223): a fresh process
                # per run, the 17 bundled fixtures in validate.c's order, each decoded once into BGRA8 / BGRX8 (tests/c_client/detex_client --oneshot)
                def client_oneshot(path, runs, mode="--oneshot"):
                    fixtures = [os.path.join(ROOT, "tests", "golden", "test-texture-%s.ktx" % n) for n in
                                ("BC1", "BC1A", "BC2", "BC3", "RGTC1", "RGTC2", "SIGNED_RGTC1", "SIGNED_RGTC2", "BPTC", "BPTC_FLOAT", "ETC1", "ETC2",
                                 "ETC2_PUNCHTHROUGH", "ETC2_EAC", "EAC_R11", "EAC_RG11", "EAC_SIGNED_R11")]
                    rows = []
                    for _ in range(runs):
                        t0 = time.perf_counter()
                        r = subprocess.run([path, mode] + fixtures, capture_output=True, text=True, timeout=120, env=clean_env)
                        wall = (time.perf_counter() - t0) * 1e3
                        for line in r.stdout.splitlines():
                            if line.startswith("oneshot files="):
                                f = dict(kv.split("=") for kv in line.split()[1:])
                                rows.append({"process_wall_ms": wall, "first_call_ms": float(f["first_call_ms"]), "fixture_sequence_ms": float(f["fixture_sequence_ms"]),
                                             "init_ms": float(f["init_ms"]), "decoded": int(f["decoded"])})
                    if not rows:
                        return None
                    med = lambda k: round(sorted(x[k] for x in rows)[len(rows) // 2], 3)   # noqa: E731
                    return {k: med(k) for k in ("process_wall_ms", "first_call_ms", "fixture_sequence_ms", "init_ms")} | {"runs": len(rows), "decoded": rows[0]["decoded"]}
                clean_env = {k: v for k, v in os.environ.items() if not k.startswith(("PYTHON", "LD_PRELOAD"))}
                one = client_oneshot(client, 10)
                if one:
                    brk = client_oneshot(client, 5, "--oneshot-breakdown")
                    one["runtime_init_ms"] = brk["init_ms"] if brk else None
                    one["after_init_first_call_ms"] = round(brk["first_call_ms"] - brk["init_ms"], 3) if brk else None
                    one["after_init_fixture_sequence_ms"] = round(brk["fixture_sequence_ms"] - brk["init_ms"], 3) if brk else None
                    one["note"] = ("medians over fresh processes; first_call_ms / fixture_sequence_ms: milliseconds from main() to the first decoded 64x64 fixture and to the "
                                   "end of the 17-fixture sequence; process_wall_ms: fork to exit, seen from this script (dynamic linking of the HIP runtime and its teardown "
                                   "included); runtime_init_ms: hipInit alone (the first runtime call of a process), after_init_*: what the library adds behind it -- "
                                   "code objects, the thread's stream and buffers, seventeen first launches")
                    small["oneshot_compiled_c_client"] = one
                    if os.path.exists(client + "_reflib"):
                        small["oneshot_reference_compiled_c_client"] = client_oneshot(client + "_reflib", 10)
            result["host_tier_small"] = small
        except Exception as e:  # noqa
            log("host_tier_small failed:", e)

    def write_formats_table(self):
        """--formats-json: every format x stream U / M / C at steady state, written to a file"""
        args, W, H = self.args, self.W, self.H
        table = {}
        for f in F.FORMATS:
            for kind in ["U", "M", "C"]:
                if kind == "M" and f.name not in ("BPTC", "BPTC_FLOAT", "BPTC_SIGNED_FLOAT"):
                    continue
                d = self.make_input(f, W // 4, H // 4, kind)
                if d is None:
                    continue
                j = self.new_job(f, W, H, d, args.layout)
                us, launches = self.steady_state_us(j)
                table["%s/%s" % (f.name, kind)] = self.roofline_of(j, us)
                log(f.name, kind, table["%s/%s" % (f.name, kind)])
                del j
                torch.cuda.empty_cache()
        os.makedirs(os.path.dirname(os.path.abspath(args.formats_json)), exist_ok=True)
        json.dump(table, open(args.formats_json, "w"), indent=1, sort_keys=True)

    def run(self):
        """-> the JSON line's dict on rank 0, None elsewhere"""
        args, multi = self.args, self.multi
        self.measure_headline()
        if multi and not args.no_extras:
            self.multi_gpu_extras()
        if self.rank != 0:
            return None
        result = self.headline_result()
        if not multi and not args.no_extras:
            self.add_roofline_references(result)
        self.add_traffic(result)
        if multi:
            self.add_collective_keys(result)
        result.update(self.extras)
        if not multi:
            if not args.no_host_tier:
                self.add_host_tier(result)
            if not args.no_extras and not args.formats_json:
                self.add_per_format(result)
                self.add_whole_images(result)
            if not args.no_extras:
                self.add_host_tier_small(result)
            if not args.no_cpu:
                result["cpu_baseline"] = cpu_baseline(self.fmt, self.data, self.W, self.H)
            if args.formats_json:
                self.write_formats_table()
        return result


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--format", default="BC1")
    ap.add_argument("--size", type=int, default=8192, help="image width = per-rank height in pixels")
    ap.add_argument("--band-height", type=int, default=0, help="per-rank band height if different from --size")
    ap.add_argument("--strong-image", type=int, default=None,
                    help="strong scaling: ONE S x S image sharded by block rows over the ranks; default 32768 (BASELINE "
                         "north_star) when N > 1, off when N == 1")
    ap.add_argument("--weak", action="store_true", help="N > 1: one --size^2 image per rank as the headline (round-1 behaviour)")
    ap.add_argument("--variant", type=int, default=0, help="A/B kernel variant (needs DETEXHIP_LIB=<make lib-ab build>)")
    ap.add_argument("--stream", default="U", choices=["U", "M", "C"])
    ap.add_argument("--layout", default="linear", choices=["linear", "tiled"],
                    help="linear = detexDecompressTextureLinear (headline); tiled = detexDecompressTextureTiled (block-major output)")
    ap.add_argument("--target", default=None, help="target pixel format for the in-kernel epilogues: BGRA8, BGRX8, RGB8, FLOAT_BGRX16 (default: native)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-settle", action="store_true", help="start the contract's W + K launches cold (no settling launches before them)")
    ap.add_argument("--no-extras", action="store_true", help="skip per_format / strong_image_32768 / weak / gather extras")
    ap.add_argument("--no-host-tier", action="store_true", help="skip the host-pointer row (tools/gpu_kernel_stats.sh: its banded calls launch the headline's kernel on eighths of the image)")
    ap.add_argument("--gather", action="store_true", help="(kept for compatibility: the gather is timed by default when N > 1)")
    ap.add_argument("--formats-json", default=None, help="also bench every format (U, M, C streams), write a table to this path")
    args = ap.parse_args()

    # The contract is ONE JSON line on stdout.  Libraries print there too (gloo announces its connections, RCCL its version banner when
    # NCCL_DEBUG is set, rocprofv3 its summary): from here on file descriptor 1 is stderr's, and the line at the end goes to the saved one.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    global torch, dist, binding, F, sharding, ol, streams
    import torch
    import torch.distributed as dist
    from detex_amd import binding, formats as F, sharding
    import oracle_lib as ol
    import streams

    bench = Bench(args)
    result = bench.run()
    if result is not None:
        sys.stdout.flush()
        real_stdout.write(json.dumps(result) + "\n")
        real_stdout.flush()
    if bench.multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
