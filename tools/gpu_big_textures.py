"""Maximum-size check: 32768-wide textures (4 GiB of pixels), bit-exact at the top, middle and bottom block rows."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from detex_amd import binding, formats as F
import oracle_lib as ol
for name, W, H in (("BC1", 32768, 32768), ("BPTC_FLOAT", 32768, 8192), ("BPTC", 32768, 16384)):
    fmt = F.BY_NAME[name]; wb, hb = W // 4, H // 4
    rng = np.random.default_rng(5)
    data = rng.integers(0, 256, size=wb * hb * fmt.block_bytes, dtype=np.uint8)
    d = torch.from_numpy(data).cuda()
    out = torch.empty(W * H * fmt.pixel_bytes, dtype=torch.uint8, device="cuda")
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    binding.decompress_linear_device(fmt, d, W, H, out=out, status=status)
    torch.cuda.synchronize()
    orc = ol.Oracle(); rows = 8
    okall = True
    for r0 in (0, hb // 2 - 3, hb - rows):
        blk = data[r0 * wb * fmt.block_bytes:(r0 + rows) * wb * fmt.block_bytes]
        _, want = orc.linear(fmt, blk, W, rows * 4)
        got = out[r0 * 4 * W * fmt.pixel_bytes:(r0 + rows) * 4 * W * fmt.pixel_bytes].cpu().numpy()
        okall &= np.array_equal(got, want)
    print(name, W, H, "GiB out %.1f" % (out.numel() / 2**30), "exact at top/middle/bottom:", okall)
    del d, out; torch.cuda.empty_cache()
