"""ctypes loader of tools/ubench/libhbmref.so (fill / copy reference kernels; measurement tooling only)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "ubench", "libhbmref.so")
PATTERNS = {0: "zeros", 1: "constant dword", 2: "random dwords", 3: "random, X half of every pixel 0 (FLOAT_RGBX16-like)",
            4: "random, constant per 16 bytes", 5: "random bytes < 64", 6: "random, every second byte 0"}
_lib = None


def load():
    global _lib
    if _lib is None:
        import torch  # noqa: F401  (torch's HIP runtime first, as detex_amd.binding does)
        lib = ctypes.CDLL(LIB_PATH)
        lib.hbmref_fill.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p]
        lib.hbmref_copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
        lib.hbmref_fill_unaligned.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_void_p]
        lib.hbmref_fill_image.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p]
        lib.hbmref_fill_rows.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p]
        _lib = lib
    return _lib


def time_us(fn, launches=30, warmup=5):
    import torch
    for _ in range(warmup):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(launches):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / launches * 1e3


def fill_GBps(nbytes, pattern=2, nontemporal=True, launches=30, buf=None):
    import torch
    lib = load()
    buf = buf if buf is not None else torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    seed = [0]

    def step():
        seed[0] += 977
        if lib.hbmref_fill(buf.data_ptr(), nbytes, pattern, int(nontemporal), seed[0], st) != 0:
            raise RuntimeError("hbmref_fill failed")
    us = time_us(step, launches)
    return nbytes / (us * 1e-6) / 1e9, us


def copy_GBps(nbytes, nontemporal=True, launches=30):
    """bytes read + bytes written per second"""
    import torch
    lib = load()
    src = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    dst = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    lib.hbmref_fill(src.data_ptr(), nbytes, 2, 1, 1, st)

    def step():
        if lib.hbmref_copy(dst.data_ptr(), src.data_ptr(), nbytes, int(nontemporal), st) != 0:
            raise RuntimeError("hbmref_copy failed")
    us = time_us(step, launches)
    return 2 * nbytes / (us * 1e-6) / 1e9, us


def fill_image_GBps(width_bytes, height, pattern=2, launches=30, pitch_bytes=0, workgroups_per_cu=0):
    """write-only fill in the decode kernels' image layout (four 4 KiB row pieces per workgroup); bytes written per second.
    workgroups_per_cu (3..7): resident workgroups capped by unused dynamic LDS, as the decode kernels' launches do"""
    import torch
    lib = load()
    lib.hbmref_set_fill_image_lds.argtypes = [ctypes.c_uint]
    lib.hbmref_set_fill_image_lds(((163840 // workgroups_per_cu) & ~2047) if workgroups_per_cu else 0)
    nbytes = width_bytes * height
    buf = torch.empty((pitch_bytes or width_bytes) * height, dtype=torch.uint8, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def step():
        if lib.hbmref_fill_image(buf.data_ptr(), width_bytes, height, pitch_bytes, pattern, 7, st) != 0:
            raise RuntimeError("hbmref_fill_image failed")
    us = time_us(step, launches)
    return nbytes / (us * 1e-6) / 1e9, us


def fill_unaligned_GBps(nbytes, offset, launches=30):
    import torch
    lib = load()
    buf = torch.empty(nbytes + 256, dtype=torch.uint8, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def step():
        if lib.hbmref_fill_unaligned(buf.data_ptr() + offset, nbytes, 7, st) != 0:
            raise RuntimeError("hbmref_fill_unaligned failed")
    us = time_us(step, launches)
    return nbytes / (us * 1e-6) / 1e9, us


def fill_rows_GBps(nbytes, rows, nontemporal, launches=30):
    import torch
    lib = load()
    buf = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def step():
        if lib.hbmref_fill_rows(buf.data_ptr(), nbytes, rows, int(nontemporal), 7, st) != 0:
            raise RuntimeError("hbmref_fill_rows failed")
    us = time_us(step, launches)
    return nbytes / (us * 1e-6) / 1e9, us
