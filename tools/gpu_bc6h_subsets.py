"""BC6H launch time on the uniform-random stream and on a stream of one-subset modes only (10-13)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from detex_amd import binding, formats as F
import oracle_lib as ol
for name in ("BPTC_FLOAT", "BPTC_SIGNED_FLOAT"):
    fmt = F.BY_NAME[name]; W = H = 8192; n = (W // 4) * (H // 4)
    base = ol.stream_u(fmt, n, seed=0xD37E5000 + 9).reshape(-1, 16).copy()
    def timeit(data, label):
        d = torch.from_numpy(np.ascontiguousarray(data.reshape(-1))).cuda()
        out = torch.empty(W * H * 8, dtype=torch.uint8, device="cuda")
        for _ in range(400): binding.decompress_linear_device(fmt, d, W, H, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(200): binding.decompress_linear_device(fmt, d, W, H, out=out)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 200 * 1e3
        rows = 16
        _, want = ol.Oracle().linear(fmt, data.reshape(-1)[:rows * (W // 4) * 16], W, rows * 4)
        print(name, label, "%.1f us" % us, "frac %.3f" % (n * 144 / (us * 1e-6) / 8e12), "exact", np.array_equal(out[:want.size].cpu().numpy(), want))
    timeit(base, "stream U")
    b2 = base.copy()
    codes = np.array([0x03, 0x07, 0x0B, 0x0F], np.uint8)          # modes 10, 11, 12, 13
    b2[:, 0] = (b2[:, 0] & 0xE0) | codes[np.arange(n) % 4]
    timeit(b2, "one-subset modes only (10-13)")
