"""Launch time of sizes that are not multiples of four: the 4-aligned interior runs on the throughput kernel (aligned or
dword-aligned row stores), the last block column / row on the per-pixel kernel (two launches per call)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from detex_amd import binding, formats as F
import oracle_lib as ol
for name, sizes in (("BC1", ((8192, 8192), (8190, 8190), (8192, 8190), (8188, 8192), (8191, 8191))), ("BPTC", ((8192, 8192), (8190, 8190))),
                    ("BPTC_FLOAT", ((8192, 8192), (8190, 8190), (8191, 8191))), ("RGTC1", ((8192, 8192), (8188, 8190), (8190, 8190))),
                    ("RGTC2", ((8192, 8192), (8190, 8190)))):
    fmt = F.BY_NAME[name]
    for (W, H) in sizes:
        wb, hb = (W + 3) // 4, (H + 3) // 4
        data = ol.stream_u(fmt, wb * hb, seed=77)
        d = torch.from_numpy(np.ascontiguousarray(data)).cuda()
        out = torch.empty(W * H * fmt.pixel_bytes, dtype=torch.uint8, device="cuda")
        for _ in range(20): binding.decompress_linear_device(fmt, d, W, H, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(100): binding.decompress_linear_device(fmt, d, W, H, out=out)
        e1.record(); torch.cuda.synchronize()
        print(name, W, H, "%.1f us" % (e0.elapsed_time(e1) / 100 * 1e3), flush=True)
