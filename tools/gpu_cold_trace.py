#!/usr/bin/env python3
"""What makes the FIRST launch after idle slow (71-111 us against 41 steady, BC1 8192^2)?  GPU box.

For idle times of 0.3 s and 3 s, each scenario starts from an idle device and times its launches one by one (a HIP event pair per launch on
the launch stream) while a thread samples the shader clock and board power from the amdgpu hwmon files:
   decode          20 decode launches                                   (what bench.py's `cold` reports)
   fill_first      ONE launch of the reference fill kernel (same store shape, no decode, another code object), then 20 decode launches
                   -> if the fill pays the penalty and the decode after it does not, the cause is the DEVICE's state (clocks / power gating /
                      memory-side wake-up), not this kernel's code or data
   tiny_first      a 64x64 decode of the same format first (loads the same kernel code, touches the same tables), then the 20 launches
                   -> separates instruction-cache / table warm-up (which it would fix) from the device state (which it would not)
   other_buffers   the 20 launches write a buffer that was never touched before (fresh allocation): page-table / TLB warm-up of the output
   wake_then_work  one tiny kernel (not waited for), 200 us of host work (the file read a client does before it decodes), then the 20
                   launches: can the wake-up be paid EARLY, hidden behind host work?  (Round 5 measured this with a library entry made for
                   the purpose, detexhipWakeDevice: the call itself took 78-780 us and the decode 200 us later still paid 92-146 us for its
                   first launch -- the device is back in its gated state within ~200 us -- so the entry was not kept:
                   profiles/r05/cold_trace_bc1_with_wake_call.json)
   spin_wait       no idle at all between measurement blocks (control)
Prints one JSON object; profiles/r05/cold_trace.json is a copy."""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch                                    # noqa: E402
import oracle_lib as ol                         # noqa: E402
from detex_amd import binding, formats as F     # noqa: E402
import hbmref                                   # noqa: E402


def hwmon_dir():
    import glob
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        if os.path.exists(os.path.join(d, "freq1_input")):
            return d
    return None


class Sampler:
    def __init__(self):
        self.dir, self.rows, self.stop = hwmon_dir(), [], threading.Event()

    def _read(self, name):
        try:
            return int(open(os.path.join(self.dir, name)).read())
        except Exception:  # noqa
            return None

    def __enter__(self):
        self.t0 = time.perf_counter()
        if self.dir:
            def poll():
                while not self.stop.is_set():
                    self.rows.append((round((time.perf_counter() - self.t0) * 1e3, 3), self._read("freq1_input"), self._read("power1_input")))
                    time.sleep(0.0005)
            self.th = threading.Thread(target=poll); self.th.start()
        return self

    def __exit__(self, *a):
        self.stop.set()
        if self.dir:
            self.th.join()

    def summary(self):
        ok = [r for r in self.rows if r[1]]
        if not ok:
            return None
        return {"first_sclk_mhz": round(ok[0][1] / 1e6), "min_sclk_mhz": round(min(r[1] for r in ok) / 1e6), "max_sclk_mhz": round(max(r[1] for r in ok) / 1e6),
                "first_power_w": None if ok[0][2] is None else round(ok[0][2] / 1e6), "samples": len(ok),
                "trace_ms_mhz_w": [(r[0], round(r[1] / 1e6), None if r[2] is None else round(r[2] / 1e6)) for r in ok[:40]]}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "BC1"
    side = 8192
    binding.load()
    fmt = F.BY_NAME[name]
    data = ol.stream_u(fmt, (side // 4) ** 2)
    d_blocks = torch.from_numpy(np.ascontiguousarray(data)).cuda()
    d_out = torch.empty(side * side * fmt.pixel_bytes, dtype=torch.uint8, device="cuda")
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    small = torch.from_numpy(np.ascontiguousarray(ol.stream_u(fmt, 256))).cuda()
    small_out = torch.empty(64 * 64 * fmt.pixel_bytes, dtype=torch.uint8, device="cuda")

    def decode(out=d_out):
        binding.decompress_linear_device(fmt, d_blocks, side, side, out=out, status=status)

    import ctypes
    fill_lib = hbmref.load()
    fill_buf = torch.empty(side * side * fmt.pixel_bytes, dtype=torch.uint8, device="cuda")

    def fill_once():
        assert fill_lib.hbmref_fill_image(fill_buf.data_ptr(), side * fmt.pixel_bytes, side, 0, 2, 7, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0

    wake = status.zero_                          # (one tiny kernel on torch's stream)

    def timed(fn, n):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        ev[0].record()
        for k in range(n):
            fn(); ev[k + 1].record()
        torch.cuda.synchronize()
        return [round(ev[k].elapsed_time(ev[k + 1]) * 1e3, 2) for k in range(n)]

    for _ in range(300):
        decode()
    for _ in range(20):
        fill_once()
    torch.cuda.synchronize()
    steady = timed(decode, 200)
    res = {"format": name, "side": side, "steady_us_median": sorted(steady)[100], "scenarios": {}}
    row_bytes = side * fmt.pixel_bytes
    for idle in (0.3, 3.0):
        for scen in ("decode", "fill_first", "tiny_first", "other_buffers", "wake_then_work", "spin_wait"):
            if scen == "spin_wait" and idle != 0.3:
                continue
            for _ in range(300):
                decode()
            torch.cuda.synchronize()
            fresh = torch.empty(side * side * fmt.pixel_bytes, dtype=torch.uint8, device="cuda") if scen == "other_buffers" else None
            torch.cuda.synchronize()
            if scen != "spin_wait":
                time.sleep(idle)
            row = {}
            with Sampler() as smp:
                if scen == "fill_first":
                    row["fill_first_launch_us"] = timed(fill_once, 1)[0]                    # ONE launch of the reference fill into its own (pre-allocated) image
                elif scen == "wake_then_work":
                    t0 = time.perf_counter()
                    wake()
                    row["wake_call_us"] = round((time.perf_counter() - t0) * 1e6, 1)
                    while time.perf_counter() - t0 < 200e-6:
                        pass
                elif scen == "tiny_first":
                    row["tiny_first_launch_us"] = timed(lambda: binding.decompress_linear_device(fmt, small, 64, 64, out=small_out, status=status), 1)[0]
                row["decode_us"] = timed((lambda: decode(fresh)) if fresh is not None else decode, 20)
            row["clocks"] = smp.summary()
            res["scenarios"]["%s/idle_%.1fs" % (scen, idle)] = row
            del fresh
    # the steady fill for reference (same process)
    for _ in range(100):
        fill_once()
    res["fill_steady_us"] = sorted(timed(fill_once, 50))[25]
    print(json.dumps(res))


if __name__ == "__main__":
    main()
