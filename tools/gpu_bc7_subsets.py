"""BC7 launch time on streams with and without three-subset modes (the wave-uniform subset trimming of decode_bptc.h)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from detex_amd import binding, formats as F
import oracle_lib as ol
fmt = F.BY_NAME["BPTC"]; W = H = 8192; n = (W // 4) * (H // 4)
base = ol.stream_u(fmt, n, seed=0xD37E5000 + 3).reshape(-1, 16).copy()
def timeit(data, label):
    d = torch.from_numpy(np.ascontiguousarray(data.reshape(-1))).cuda()
    out = torch.empty(W * H * 4, dtype=torch.uint8, device="cuda")
    for _ in range(400): binding.decompress_linear_device(fmt, d, W, H, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(200): binding.decompress_linear_device(fmt, d, W, H, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 200 * 1e3
    # verify a slice
    orc = ol.Oracle(); rows = 16
    _, want = orc.linear(fmt, data.reshape(-1)[:rows * (W // 4) * 16], W, rows * 4)
    okv = np.array_equal(out[:want.size].cpu().numpy(), want)
    print(label, "%.1f us" % us, "frac %.3f" % (n * 80 / (us * 1e-6) / 8e12), "exact", okv)
timeit(base, "stream U (all modes, 3-subset blocks in every wave)")
b2 = base.copy()
low = b2[:, 0]
m0 = (low & 1) == 1                       # mode 0 -> mode 6 (bit 6)
b2[m0, 0] = (low[m0] & 0x80) | 0x40
m2 = ((b2[:, 0] & 7) == 4)                # mode 2 -> mode 1 (bit 1)
b2[m2, 0] = (b2[m2, 0] & 0xFC) | 0x02
timeit(b2, "no three-subset modes (0 -> 6, 2 -> 1)")
b3 = base.copy(); b3[:, 0] = (b3[:, 0] & 0x80) | 0x40      # all mode 6
timeit(b3, "mode 6 only")
b4 = base.copy()
hi_modes = ((b4[:, 0] & 0x0F) == 0) & (b4[:, 0] != 0)      # modes 4-7 -> mode 3
b4[hi_modes, 0] = (b4[hi_modes, 0] & 0xF0) | 0x08
timeit(b4, "opaque modes only (4-7 -> 3)")
