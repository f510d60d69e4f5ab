"""Reference points for the HBM roofline on this box: what plain fill / copy kernels of the same footprint reach.
(bytes moved per second; copy counts read + write)"""
import torch
def t(fn, n=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mib in (256, 1024):
    nbytes = mib << 20
    a = torch.empty(nbytes // 4, dtype=torch.int32, device="cuda"); b = torch.empty_like(a)
    s = t(lambda: a.fill_(7)); print("fill  %4d MiB: %7.1f us  %.2f TB/s written" % (mib, s * 1e6, nbytes / s / 1e12))
    s = t(lambda: torch.cuda.memset if False else a.zero_()); print("zero  %4d MiB: %7.1f us  %.2f TB/s written" % (mib, s * 1e6, nbytes / s / 1e12))
    s = t(lambda: b.copy_(a)); print("copy  %4d MiB: %7.1f us  %.2f TB/s read+written" % (mib, s * 1e6, 2 * nbytes / s / 1e12))
# hipMemsetD32Async with a zero and a non-zero pattern (is the fast "zero" a tuned kernel or a property of zeros?)
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetD32Async.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
for mib in (256, 1024):
    nbytes = mib << 20
    a = torch.empty(nbytes // 4, dtype=torch.int32, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for val in (0, 0x12345678):
        s = t(lambda: hip.hipMemsetD32Async(a.data_ptr(), val, nbytes // 4, st))
        print("hipMemsetD32Async(0x%08X) %4d MiB: %7.1f us  %.2f TB/s written" % (val, mib, s * 1e6, nbytes / s / 1e12))
