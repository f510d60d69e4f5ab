"""Launch time per 25-launch window over a long run: shows clock/power behaviour of VALU-heavy kernels.
usage: python tools/gpu_sustain.py FORMAT [windows]"""
import sys, os, json, subprocess, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from detex_amd import binding, formats as F
import oracle_lib as ol

name = sys.argv[1]; windows = int(sys.argv[2]) if len(sys.argv) > 2 else 40
fmt = F.BY_NAME[name]; W = H = 8192
data = ol.stream_u(fmt, (W // 4) * (H // 4), seed=0xD37E5000 + fmt.index)
d_blocks = torch.from_numpy(np.ascontiguousarray(data)).cuda()
d_out = torch.empty(W * H * fmt.pixel_bytes, dtype=torch.uint8, device="cuda")
clocks = []
stop = False
def poll():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
            j = json.loads(o); c = j[sorted(j)[0]]
            clocks.append({k: v for k, v in c.items() if "sclk" in k or "mclk" in k or "ower" in k})
        except Exception as e:  # noqa
            clocks.append(str(e)[:80])
        time.sleep(0.02)
t = threading.Thread(target=poll); t.start()
for _ in range(3): binding.decompress_linear_device(fmt, d_blocks, W, H, out=d_out)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(windows + 1)]
ev[0].record()
for w in range(windows):
    for _ in range(25): binding.decompress_linear_device(fmt, d_blocks, W, H, out=d_out)
    ev[w + 1].record()
torch.cuda.synchronize()
stop = True; t.join()
print(name, "us per launch per 25-launch window:", [round(ev[i].elapsed_time(ev[i + 1]) / 25 * 1e3, 1) for i in range(windows)])
print("clock samples:", clocks[:3], "...", clocks[-3:], len(clocks))
