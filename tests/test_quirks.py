"""The quirk switch (detexhipSetQuirks / DETEXHIP_QUIRKS): the reference's two deviations from the BPTC specification
 are reproduced by default and can be switched off per thread.

CPU part: the checker's own switch (oracle/detex_oracle.c: orc_set_quirks) is pinned -- with the affected input bit clear
both settings equal the compiled reference, with it set they differ; BC7 mode 6 with the quirk off equals an independent
decoder written here from the format definition.  GPU part (-m gpu): the library with quirks off equals the checker with
quirks off through every entry point, the setting is per thread, and the default is unchanged."""
import ctypes
import threading

import numpy as np
import pytest

import oracle_lib as ol
import streams
from detex_amd import formats as F

BC7 = F.BY_NAME["BPTC"]
BC6H = [F.BY_NAME["BPTC_FLOAT"], F.BY_NAME["BPTC_SIGNED_FLOAT"]]
Q_BC7, Q_BC6H, Q_ALL = 1, 2, 3


def _mode6_blocks(n, seed):
    b = ol.stream_u(BC7, n, seed=seed).reshape(-1, 16).copy()
    b[:, 0] = (b[:, 0] & 0x80) | 0x40                      # mode 6: 0000001 then the fields
    return b


def _mode12_blocks(fmt, n, seed):
    b = ol.stream_u(fmt, n, seed=seed).reshape(-1, 16).copy()
    b[:, 0] = (b[:, 0] & 0xE0) | streams.BC6H_MODE_CODES[12]
    return b


def _set(orc, mask):
    orc.lib.orc_set_quirks.argtypes = [ctypes.c_uint]
    orc.lib.orc_set_quirks(mask)


def _bc7_mode6_spec(block):
    """BC7 mode 6 from the format definition: 7-bit mode field, RGBA endpoints of 7 bits in the order R0 R1 G0 G1 B0 B1 A0 A1,
    one P-bit per endpoint, sixteen 4-bit indices (the first one 3 bits: anchor), LSB first."""
    bits = int.from_bytes(bytes(block), "little")
    pos = [7]

    def take(n):
        v = (bits >> pos[0]) & ((1 << n) - 1)
        pos[0] += n
        return v
    comp = [[take(7), take(7)] for _ in range(4)]           # [channel][endpoint]
    p = [take(1), take(1)]
    ep = [[(comp[c][e] << 1) | p[e] for c in range(4)] for e in range(2)]
    weights = [0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64]
    out = np.zeros(64, np.uint8)
    for t in range(16):
        w = weights[take(3 if t == 0 else 4)]
        for c in range(4):
            out[4 * t + c] = ((64 - w) * ep[0][c] + w * ep[1][c] + 32) >> 6
    return out


def test_oracle_switch_is_pinned(oracle, ref):
    try:
        # A-2: BC7 mode 6
        b = _mode6_blocks(4096, 11)
        _set(oracle, Q_ALL)
        _, on = oracle.blocks(BC7, b)
        _set(oracle, Q_ALL & ~Q_BC7)
        _, off = oracle.blocks(BC7, b)
        bit64 = (b[:, 8] & 1).astype(bool)
        assert np.array_equal(on[~bit64], off[~bit64]), "with block bit 64 clear the switch must not matter"
        assert (on[bit64] != off[bit64]).any(axis=1).mean() > 0.9, "with block bit 64 set nearly every block must change"
        for i in range(0, 4096, 37):
            assert np.array_equal(off[i], _bc7_mode6_spec(b[i])), i
        if ref is not None:
            fn = ref.block_fn(BC7)
            for i in range(0, 4096, 53):
                o = np.zeros(64, np.uint8)
                assert fn(ol._ptr(b[i]), 0xFFFFFFFF, 0, ol._ptr(o)) and np.array_equal(o, on[i])
        # other modes are untouched by the BC7 switch, and the BC6H switch does not touch BC7
        u = ol.stream_u(BC7, 4096, seed=3).reshape(-1, 16)
        u = u[(u[:, 0] & 0x7F) != 0x40]
        _set(oracle, Q_ALL)
        _, a = oracle.blocks(BC7, u)
        _set(oracle, 0)
        _, z = oracle.blocks(BC7, u)
        assert np.array_equal(a, z)
        # A-3: BC6H mode 12
        for fmt in BC6H:
            b = _mode12_blocks(fmt, 4096, 17 + fmt.index)
            _set(oracle, Q_ALL)
            _, on = oracle.blocks(fmt, b)
            _set(oracle, Q_ALL & ~Q_BC6H)
            _, off = oracle.blocks(fmt, b)
            bit63 = (b[:, 7] & 0x80).astype(bool)
            assert np.array_equal(on[~bit63], off[~bit63])
            assert (on[bit63] != off[bit63]).any(axis=1).mean() > 0.5
            # what the switch restores is ONE bit of one endpoint component: clearing the bit in the input gives the quirk's output
            cleared = b.copy()
            cleared[:, 7] &= 0x7F
            _, off_cleared = oracle.blocks(fmt, cleared)
            assert np.array_equal(off_cleared, on)
            u = streams.stream_m(fmt, ol.stream_u(fmt, 4096, seed=5)).reshape(-1, 16)
            u = u[(u[:, 0] & 0x1F) != streams.BC6H_MODE_CODES[12]]
            _set(oracle, Q_ALL)
            _, a = oracle.blocks(fmt, u)
            _set(oracle, 0)
            _, z = oracle.blocks(fmt, u)
            assert np.array_equal(a, z)
    finally:
        _set(oracle, Q_ALL)


@pytest.fixture
def gpu_lib():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from detex_amd import binding
    lib = binding.load()
    lib.detexhipSetQuirks.argtypes = [ctypes.c_uint32]
    lib.detexhipGetQuirks.restype = ctypes.c_uint32
    yield lib
    lib.detexhipSetQuirks(Q_ALL)


@pytest.mark.gpu
@pytest.mark.parametrize("quirks", [0, Q_BC7, Q_BC6H, Q_ALL])
def test_library_switch_matches_checker(quirks, gpu_lib, oracle, hiplib):
    import torch
    from detex_amd import binding
    try:
        gpu_lib.detexhipSetQuirks(quirks)
        assert gpu_lib.detexhipGetQuirks() == quirks
        _set(oracle, quirks)
        cases = [(BC7, np.concatenate([_mode6_blocks(1024, 21), ol.stream_u(BC7, 3072, seed=22).reshape(-1, 16)]))]
        for fmt in BC6H:
            cases.append((fmt, np.concatenate([_mode12_blocks(fmt, 1024, 23), streams.stream_m(fmt, ol.stream_u(fmt, 3072, seed=24)).reshape(-1, 16)])))
        for fmt, blocks in cases:
            data = np.ascontiguousarray(blocks).reshape(-1)
            dev = torch.from_numpy(data).cuda()
            n = len(blocks)
            ok_o, want = oracle.blocks(fmt, blocks)
            got, ok = binding.decompress_blocks_device(fmt, dev, n)                     # per-block API (checked kernels)
            torch.cuda.synchronize()
            assert np.array_equal(got.cpu().numpy().reshape(n, -1), want) and np.array_equal(ok.cpu().numpy()[:n].astype(bool), ok_o)
            W, H = 256, n // 64 * 4                                                     # linear: throughput kernel (uniform waves of mode 6 / 12 first)
            _, want_l = oracle.linear(fmt, data, W, H)
            got_l = binding.decompress_linear_device(fmt, dev, W, H)
            _, want_t = oracle.tiled(fmt, data, W // 4, H // 4)
            got_t = binding.decompress_tiled_device(fmt, dev, W // 4, H // 4)
            torch.cuda.synchronize()
            assert np.array_equal(got_l.cpu().numpy(), want_l) and np.array_equal(got_t.cpu().numpy(), want_t)
            _, want_c = oracle.linear(fmt, data, 250, 61)                               # clipped geometry (staged kernel)
            got_c = binding.decompress_linear_device(fmt, dev, 250, 61)
            torch.cuda.synchronize()
            assert np.array_equal(got_c.cpu().numpy(), want_c)
            ok_h, got_h = hiplib.linear(fmt, data, W, H)                                # host tier, and the one-block leaf function
            assert np.array_equal(got_h, want_l)
            # the multi-device entries: the host-image one decodes in worker threads, which must apply the CALLER's mask (one shard runs
            # inline, two and three in workers); the device one in the calling thread
            ndev = gpu_lib.detexhipGetDeviceCount()
            for shards in (1, 2, 3):
                _, got_m, _ = binding.decompress_linear_multi_device_host(fmt, data, W, H, [g % ndev for g in range(shards)])
                assert np.array_equal(got_m, want_l), (fmt.name, quirks, shards)
            r = binding.decompress_linear_multi_device(fmt, W, H, [g % ndev for g in range(2)], host_blocks=data, gather_device=0)
            assert np.array_equal(r["gathered"].cpu().numpy(), want_l)
            for i in (0, 5, 1023, 1500):
                ok_b, px = hiplib.block(fmt, blocks[i])
                assert ok_b == ok_o[i] and (not ok_b or np.array_equal(px, want[i]))
    finally:
        _set(oracle, Q_ALL)


@pytest.mark.gpu
def test_switch_is_per_thread_and_defaults_to_the_reference(gpu_lib, oracle):
    import torch
    from detex_amd import binding
    blocks = _mode6_blocks(512, 31)
    blocks[:, 8] |= 1                                           # block bit 64 set: the quirk matters for every block
    data = blocks.reshape(-1)
    _set(oracle, Q_ALL)
    _, ref_like = oracle.blocks(BC7, blocks)
    _set(oracle, 0)
    _, spec = oracle.blocks(BC7, blocks)
    _set(oracle, Q_ALL)
    assert (ref_like != spec).any()
    seen = {}

    def worker():
        lib = binding.load()
        lib.detexhipGetQuirks.restype = ctypes.c_uint32
        seen["default"] = lib.detexhipGetQuirks()
        got, _ = binding.decompress_blocks_device(BC7, torch.from_numpy(data).cuda(), 512)
        torch.cuda.synchronize()
        seen["other_thread"] = got.cpu().numpy().reshape(512, -1)
    gpu_lib.detexhipSetQuirks(0)
    t = threading.Thread(target=worker)
    t.start(); t.join()
    got, _ = binding.decompress_blocks_device(BC7, torch.from_numpy(data).cuda(), 512)
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy().reshape(512, -1), spec)
    assert seen["default"] == Q_ALL and np.array_equal(seen["other_thread"], ref_like)
