"""Block-mode histogram (detexhipModeHistogramDevice) over a side x side block stream (default 2048), timed by hipGraph replay.
usage: python tools/bench_histogram.py [side_in_blocks]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from detex_amd import binding, formats as F
import oracle_lib as ol
for name in ("BPTC", "BC1", "ETC2", "BPTC_FLOAT"):
    side_blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    fmt = F.BY_NAME[name]; n = side_blocks * side_blocks
    data = ol.stream_u(fmt, n, seed=5 + fmt.index)
    d = torch.from_numpy(np.ascontiguousarray(data)).cuda()
    h = torch.zeros(16, dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()              # the first launch on a stream allocates its scratch area: do it before the capture, on the capture stream
    with torch.cuda.stream(side):
        for _ in range(5): binding.mode_histogram_device(fmt, d, n, hist=h)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()              # 20 calls per graph: the call is ~15 us of GPU work, less than its CPU launch cost
    with torch.cuda.graph(g, stream=side):
        for _ in range(20): binding.mode_histogram_device(fmt, d, n, hist=h)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 100 * 1e3
    ga = torch.cuda.CUDAGraph()             # the accumulating entry: the kernel alone, no memset node
    with torch.cuda.graph(ga, stream=side):
        for _ in range(20): binding.mode_histogram_device(fmt, d, n, hist=h, accumulate=True)
    ga.replay(); torch.cuda.synchronize()
    e0.record()
    for _ in range(5): ga.replay()
    e1.record(); torch.cuda.synchronize()
    us_acc = e0.elapsed_time(e1) / 100 * 1e3
    binding.mode_histogram_device(fmt, d, n, hist=h); torch.cuda.synchronize()
    print(name, "histogram of %d blocks (graph replay): %.1f us, %.2f TB/s read; accumulating entry %.1f us, %.2f TB/s"
          % (n, us, n * fmt.block_bytes / us / 1e6, us_acc, n * fmt.block_bytes / us_acc / 1e6), h.cpu().numpy()[:9])
