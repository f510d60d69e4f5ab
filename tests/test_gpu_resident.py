"""GPU tests of the resident service behind the host tier's smallest calls (include/detexhip.h: detexhipSetResidentIdleMicroseconds;
detex_amd/csrc/kernels_resident.h): from the second call in a row of one (format, target) pair on, one-block calls and linear
textures of up to 1024 blocks are posted to a kernel that is already running.  Bit-exact against the oracle like every other path,
and the tests check through detexhipGetResidentStats that the resident kernel really answered."""
import ctypes
import threading
import time

import numpy as np
import pytest

import oracle_lib as ol
from detex_amd import formats as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def _stats(lib):
    lib.detexhipGetResidentStats.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(ctypes.c_ulonglong)]
    lib.detexhipGetResidentStats.restype = None
    a, b = ctypes.c_ulonglong(0), ctypes.c_ulonglong(0)
    lib.detexhipGetResidentStats(ctypes.byref(a), ctypes.byref(b))
    return a.value, b.value


def _idle(lib, us):
    lib.detexhipSetResidentIdleMicroseconds.argtypes = [ctypes.c_int]
    lib.detexhipSetResidentIdleMicroseconds.restype = ctypes.c_int
    return lib.detexhipSetResidentIdleMicroseconds(us)


@pytest.fixture()
def resident(hiplib, torch_cuda):
    """the service with a generous idle time (the tests below are Python-paced); the default is restored afterwards"""
    before = _idle(hiplib.lib, 20000)
    yield hiplib
    _idle(hiplib.lib, before)


@pytest.mark.parametrize("fmt", F.FORMATS, ids=lambda f: f.name)
def test_one_block_calls_every_format(fmt, resident, oracle):
    """the 19 leaf functions (detex.h:435-531), eight calls in a row each: call 1 is a launch, calls 2-8 are answered by the
    resident kernel; random blocks incl. invalid ones (reserved modes), each == the oracle's pixels and bool"""
    lib = resident
    blocks = ol.stream_u(fmt, 8, seed=0xB10C + fmt.index).reshape(8, fmt.block_bytes)
    served0, _ = _stats(lib.lib)
    for k in range(8):
        ok, got = lib.block(fmt, blocks[k])
        want_ok, want = oracle.block(fmt, blocks[k])
        assert ok == want_ok, (fmt.name, k)
        if ok:
            assert np.array_equal(got, want), (fmt.name, k)
    served1, _ = _stats(lib.lib)
    assert served1 - served0 in (7, 8)          # (8: the previous test's last small call was this pair already)


GEOMETRIES = [(4, 4), (64, 64), (128, 128), (100, 36), (7, 13), (256, 16), (16, 252), (124, 128)]


@pytest.mark.parametrize("name", ["BC1", "BC3", "RGTC1", "SIGNED_RGTC2", "ETC2", "ETC2_EAC", "EAC_RG11", "BPTC", "BPTC_FLOAT", "BPTC_SIGNED_FLOAT"])
def test_small_linear_textures(name, resident, oracle):
    """detexDecompressTextureLinear (texture.c:105-145) on textures of 1 to 1024 blocks, clipped sizes included, twice each with
    different data: one tile, several tiles (the other workgroups are woken), 64-bit pixels through the LDS transpose"""
    lib = resident
    fmt = F.BY_NAME[name]
    served0, _ = _stats(lib.lib)
    calls = 0
    for rep in range(2):
        for (W, H) in GEOMETRIES:
            wb, hb = (W + 3) // 4, (H + 3) // 4
            data = ol.stream_u(fmt, wb * hb, seed=0x5EED + 97 * rep + W * 131 + H)
            ok, got = lib.linear(fmt, data, W, H)
            want_ok, want = oracle.linear(fmt, data, W, H)
            assert np.array_equal(got, want), (name, W, H, rep)
            assert ok == want_ok
            calls += 1
    served1, _ = _stats(lib.lib)
    assert served1 - served0 in (calls - 1, calls)     # (calls: the pair was already the previous test's)


@pytest.mark.parametrize("name,target", [("BC1", "BGRA8"), ("BC3", "RGB8"), ("BPTC_FLOAT", "FLOAT_BGRX16"), ("RGTC2", "RGBX8"), ("EAC_R11", "RGB8")])
def test_small_textures_with_target_formats(name, target, resident, oracle):
    lib = resident
    fmt = F.BY_NAME[name]
    pf = getattr(F, "PIXEL_FORMAT_" + target)
    for (W, H) in [(64, 64), (36, 20), (128, 96)]:
        wb, hb = (W + 3) // 4, (H + 3) // 4
        for rep in range(2):
            data = ol.stream_u(fmt, wb * hb, seed=0x7A6 + rep + W)
            ok, got = lib.linear(fmt, data, W, H, pixel_format=pf)
            want_ok, want = oracle.linear_to(fmt, data, W, H, pf)
            assert np.array_equal(got, want), (name, target, W, H)
            assert ok == want_ok


@pytest.mark.parametrize("name", ["BC1", "BC2", "RGTC1", "ETC2_EAC", "EAC_SIGNED_RG11", "BPTC", "BPTC_FLOAT"])
def test_small_block_major_textures(name, resident, oracle):
    """detexDecompressTextureTiled (texture.c:77-98) on 1 to 1024 blocks, three calls per size: one tile (tagged chunks) and several"""
    lib = resident
    fmt = F.BY_NAME[name]
    served0, _ = _stats(lib.lib)
    calls = 0
    for (wb, hb) in [(1, 1), (16, 16), (5, 3), (32, 32), (64, 9), (16, 17)]:
        for rep in range(3):
            data = ol.stream_u(fmt, wb * hb, seed=0x71ED + 13 * rep + wb * 37 + hb)
            ok, got = lib.tiled(fmt, data, wb, hb)
            want_ok, want = oracle.tiled(fmt, data, wb, hb)
            assert np.array_equal(got, want), (name, wb, hb, rep)
            assert ok == want_ok
            calls += 1
    served1, _ = _stats(lib.lib)
    assert served1 - served0 in (calls - 1, calls)


def test_linear_and_block_major_requests_share_a_kernel(resident, oracle):
    lib = resident
    fmt = F.BY_NAME["BC3"]
    _, started0 = _stats(lib.lib)
    for k in range(8):
        data = ol.stream_u(fmt, 256, seed=40 + k)
        if k % 2:
            ok, got = lib.tiled(fmt, data, 16, 16)
            assert np.array_equal(got, oracle.tiled(fmt, data, 16, 16)[1])
        else:
            ok, got = lib.linear(fmt, data, 64, 64)
            assert np.array_equal(got, oracle.linear(fmt, data, 64, 64)[1])
    _, started1 = _stats(lib.lib)
    assert started1 - started0 <= 1


def test_format_switches_and_block_texture_mix(resident, oracle):
    """pairs alternate: a resident kernel of one pair must never answer a request meant for another (it is stopped first), and
    one-block and texture requests of one pair share a kernel"""
    lib = resident
    seq = ["BC1", "BC1", "BPTC", "BPTC", "BPTC", "BC1", "ETC2", "ETC2", "BC1", "BC1", "BC1", "BPTC_FLOAT", "BPTC_FLOAT"]
    for k, name in enumerate(seq):
        fmt = F.BY_NAME[name]
        if k % 3 == 2:
            blk = ol.stream_u(fmt, 1, seed=900 + k)
            ok, got = lib.block(fmt, blk)
            want_ok, want = oracle.block(fmt, blk)
            assert ok == want_ok
            if ok:
                assert np.array_equal(got, want), (k, name)
        else:
            data = ol.stream_u(fmt, 16 * 16, seed=800 + k)
            ok, got = lib.linear(fmt, data, 64, 64)
            want_ok, want = oracle.linear(fmt, data, 64, 64)
            assert np.array_equal(got, want), (k, name)
            assert ok == want_ok


def test_idle_exit_and_relaunch(hiplib, torch_cuda, oracle):
    """the kernel leaves after the idle time; the next request finds it gone and starts a new instance; a device-wide
    synchronisation returns (nothing resident outlives its idle time)"""
    torch = torch_cuda
    lib = hiplib
    before = _idle(lib.lib, 200)
    try:
        fmt = F.BY_NAME["BC2"]
        for round_ in range(3):
            _, started0 = _stats(lib.lib)
            for k in range(3):
                data = ol.stream_u(fmt, 256, seed=31 * round_ + k)
                ok, got = lib.linear(fmt, data, 64, 64)
                assert np.array_equal(got, oracle.linear(fmt, data, 64, 64)[1])
            _, started1 = _stats(lib.lib)
            assert started1 > started0                  # Python-paced calls: at least one new instance per round
            time.sleep(0.01)
            t0 = time.perf_counter()
            torch.cuda.synchronize()
            assert time.perf_counter() - t0 < 0.5
    finally:
        _idle(lib.lib, before)


def test_service_switched_off(hiplib, torch_cuda, oracle):
    lib = hiplib
    before = _idle(lib.lib, 0)
    try:
        fmt = F.BY_NAME["ETC1"]
        served0, started0 = _stats(lib.lib)
        for k in range(4):
            data = ol.stream_u(fmt, 64, seed=k)
            ok, got = lib.linear(fmt, data, 32, 32)
            assert np.array_equal(got, oracle.linear(fmt, data, 32, 32)[1])
        assert _stats(lib.lib) == (served0, started0)
    finally:
        _idle(lib.lib, before)


def test_concurrent_threads_each_with_their_own_kernel(resident, oracle):
    lib = resident
    errors = []

    def worker(t):
        try:
            fmt = F.BY_NAME[["BC1", "BPTC", "ETC2_EAC", "BPTC_FLOAT"][t % 4]]
            for k in range(12):
                data = ol.stream_u(fmt, 256, seed=1000 * t + k)
                ok, got = lib.linear(fmt, data, 64, 64)
                if not np.array_equal(got, oracle.linear(fmt, data, 64, 64)[1]):
                    errors.append((t, k))
            served, started = _stats(lib.lib)
            if served < 11:
                errors.append((t, "served", served))
            lib.lib.detexhipReleaseThreadResources.restype = None
            lib.lib.detexhipReleaseThreadResources()
        except Exception as e:  # noqa
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


def test_release_with_a_resident_kernel_running(resident, torch_cuda, oracle):
    """detexhipReleaseThreadResources while an instance lingers: it is stopped, the buffers go, the next call starts over"""
    lib = resident
    fmt = F.BY_NAME["BC1"]
    for round_ in range(2):
        for k in range(3):
            data = ol.stream_u(fmt, 256, seed=5 + k)
            assert np.array_equal(lib.linear(fmt, data, 64, 64)[1], oracle.linear(fmt, data, 64, 64)[1])
        lib.lib.detexhipReleaseThreadResources.restype = None
        lib.lib.detexhipReleaseThreadResources()
    torch_cuda.cuda.synchronize()


def test_multi_tile_requests_with_changing_payloads(resident, oracle):
    """requests for more than one tile (the leader hands them to the other workgroups through device memory) back to back, each with
    ANOTHER geometry / layout than the one before -- 2, 3 and 4 tiles, clipped and whole, linear and block-major: every workgroup
    acknowledges every such request before `done` is published (kernels_resident.h: resident_acknowledge), so no follower can pair a
    stale request number with a newer payload; 600 calls, each == the oracle"""
    lib = resident
    fmt = F.BY_NAME["BC3"]
    shapes = [(128, 128, False), (96, 64, False), (125, 90, False), (128, 64, True), (68, 68, False), (112, 96, True), (128, 124, False)]
    cases = []
    for k, (w, h, tiled) in enumerate(shapes):
        wb, hb = (w + 3) // 4, (h + 3) // 4
        assert 256 < wb * hb <= 1024
        data = ol.stream_u(fmt, wb * hb, seed=0x7117 + k)
        want = oracle.tiled(fmt, data, wb, hb) if tiled else oracle.linear(fmt, data, w, h)
        cases.append((w, h, wb, hb, tiled, data, want))
    served0, _ = _stats(lib.lib)
    rng = np.random.default_rng(5)
    for k in range(600):
        w, h, wb, hb, tiled, data, (want_ok, want) = cases[int(rng.integers(len(cases)))]
        ok, got = lib.tiled(fmt, data, wb, hb) if tiled else lib.linear(fmt, data, w, h)
        assert ok == want_ok and np.array_equal(got, want), (k, w, h, tiled)
    served1, _ = _stats(lib.lib)
    assert served1 - served0 >= 590
