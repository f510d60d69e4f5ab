"""Seeded randomised parity cases shared by tests/test_gpu_parity.py::test_seeded_fuzz_slice (a fixed slice in `-m gpu`) and
tools/gpu_fuzz.py (open-ended sweep on the GPU box).  One seed = every format once: a random geometry (power-of-two and
non-power-of-two block widths, both sides of the several-blocks-per-lane condition, clipped sizes, a padded pitch now and
then), a stream biased towards repeated blocks one time in three, decoded linear (native target and, one time in three, a
random epilogue target), block-major and through the per-block API with a random mode mask (device pointers, and a random count
of the blocks through the batched host-pointer entry detexhipDecompressBlocks) -- and, for textures of up to 1024
blocks, through the HOST-POINTER entry points twice in a row (the second call is answered by the resident kernel) plus two one-block
leaf calls with a random mode mask -- each against the CPU oracle, bit for bit; and every 29th seed one larger texture through the
host-pointer tier's banded / staged / duplex paths against the device tier (_large_host_call).  Test infrastructure only."""
import numpy as np

from detex_amd import formats as F
import oracle_lib as ol

GEOMETRIES = [(256, 64), (1024, 128), (72, 40), (100, 36), (4, 4), (260, 12), (1028, 8), (62, 30), (513, 17), (8000, 16), (1000, 52),
              (2004, 24), (4093, 9), (36, 256), (12, 1024), (8192, 8), (1, 1), (3, 7), (4100, 4)]


_API = []


def _host_api(binding):
    if not _API:
        _API.append(ol.DetexAPI(binding.LIB_PATH))
    return _API[0]


def run_seed(seed, oracle, binding, torch):
    """returns the number of decode calls checked"""
    rng = np.random.default_rng(seed)
    cases = 0
    api = _host_api(binding)
    for fmt in F.FORMATS:
        W, H = GEOMETRIES[int(rng.integers(0, len(GEOMETRIES)))]
        wb, hb = (W + 3) // 4, (H + 3) // 4
        data = ol.stream_u(fmt, wb * hb, seed=int(rng.integers(1, 1 << 40)))
        if rng.integers(0, 3) == 0:     # uniform waves, repeated rare-mode blocks
            blk = data.reshape(-1, fmt.block_bytes)
            src = blk[int(rng.integers(0, len(blk)))].copy()
            blk[rng.integers(0, len(blk), max(1, len(blk) // int(rng.integers(2, 40))))] = src
        dev = torch.from_numpy(np.ascontiguousarray(data)).cuda()
        where = (fmt.name, W, H, seed)
        # linear, native target, now and then into a padded pitch with a canary
        px = fmt.pixel_bytes
        pad = int(rng.integers(0, 4)) * (4 if px >= 4 else px) if rng.integers(0, 4) == 0 else 0
        pitch = W * px + pad
        ok_o, want = oracle.linear(fmt, data, W, H)
        status = torch.zeros(1, dtype=torch.int32, device="cuda")
        canvas = torch.full((H * pitch + 64,), 0xA5, dtype=torch.uint8, device="cuda")
        binding.decompress_linear_device(fmt, dev, W, H, out=canvas, pitch=pitch, status=status)
        torch.cuda.synchronize()
        got = canvas.cpu().numpy()
        img = got[:H * pitch].reshape(H, pitch)
        assert np.array_equal(img[:, :W * px].reshape(-1), np.asarray(want).reshape(-1)), ("linear",) + where + (pitch,)
        assert (img[:, W * px:] == 0xA5).all() and (got[H * pitch:] == 0xA5).all(), ("linear wrote outside the image",) + where + (pitch,)
        assert bool(status.item() == 0) == ok_o, ("status",) + where
        cases += 1
        # linear, an epilogue target
        targets = [pf for pf in F.accepted_pixel_formats(fmt) if F.epilogue_kind(fmt, pf)]
        if targets and rng.integers(0, 3) == 0:
            pf = targets[int(rng.integers(0, len(targets)))]
            _, want_pf = oracle.linear_to(fmt, data, W, H, pf)
            got_pf = binding.decompress_linear_device(fmt, dev, W, H, pixel_format=pf)
            torch.cuda.synchronize()
            assert np.array_equal(got_pf.cpu().numpy().reshape(-1), np.asarray(want_pf).reshape(-1)), ("linear target 0x%04X" % pf,) + where
            cases += 1
        # block-major
        _, want_t = oracle.tiled(fmt, data, wb, hb)
        got_t = binding.decompress_tiled_device(fmt, dev, wb, hb)
        torch.cuda.synchronize()
        assert np.array_equal(got_t.cpu().numpy(), want_t), ("tiled",) + where
        # per-block API with a random mode mask
        mask = int(rng.integers(0, 1 << 14)) | (0 if rng.integers(0, 2) else 0xFFFFFFFF)
        ok_b, want_b = oracle.blocks(fmt, data, mode_mask=mask)
        got_b, got_ok = binding.decompress_blocks_device(fmt, dev, wb * hb, mode_mask=mask)
        torch.cuda.synchronize()
        assert np.array_equal(got_ok.cpu().numpy()[:wb * hb].astype(bool), ok_b), ("blocks ok", hex(mask)) + where
        assert np.array_equal(got_b.cpu().numpy().reshape(-1)[:wb * hb * 16 * px], want_b.reshape(-1)), ("blocks", hex(mask)) + where
        cases += 2
        # the batched HOST-pointer block entry with the same mask (pinned exchange or staged, by size; a random count of the blocks)
        n_h = int(rng.integers(1, wb * hb + 1))
        all_ok, ok_h, px_h = api.blocks(fmt, data[:n_h * fmt.block_bytes], mode_mask=mask)
        assert np.array_equal(ok_h.astype(bool), ok_b[:n_h]) and all_ok == bool(ok_b[:n_h].all()), ("host blocks ok", hex(mask), n_h) + where
        assert np.array_equal(px_h, want_b[:n_h]), ("host blocks", hex(mask), n_h) + where
        cases += 1
        # host-pointer tier, small textures: two calls in a row (launch, then the resident kernel), either layout; two leaf calls
        if wb * hb <= 1024:
            for rep in range(2):
                if rng.integers(0, 3) == 0:
                    ok_h, got_h = api.tiled(fmt, data, wb, hb)
                    assert np.array_equal(got_h, want_t), ("host tiled", rep) + where
                else:
                    ok_h, got_h = api.linear(fmt, data, W, H)
                    assert np.array_equal(got_h, np.asarray(want).reshape(-1)), ("host linear", rep) + where
                assert ok_h == ok_o, ("host ok", rep) + where
            blk = data.reshape(-1, fmt.block_bytes)
            for rep in range(2):
                k = int(rng.integers(0, len(blk)))
                ok_1, got_1 = api.block(fmt, blk[k], mode_mask=mask)
                assert ok_1 == bool(ok_b[k]), ("host block ok", hex(mask)) + where
                if ok_1:
                    assert np.array_equal(got_1, want_b[k]), ("host block", hex(mask)) + where
            cases += 4
    cases += _large_host_call(seed, rng, api, binding, torch)
    return cases


# The host-pointer tier's larger paths (host_tier.cpp): every 29th seed a texture of the pinned exchange's banded range or of the staged path
# (status word in pinned memory / in device memory), every 499th one with 32+ MiB of blocks (uploaded beside its download by the library's
# helper thread) -- against the DEVICE tier on the same blocks (which the rest of this file holds against the oracle), either layout.
LARGE_GEOMETRIES = [(512, 512), (724, 640), (1024, 1024), (2048, 1024), (2048, 2052), (1001, 513), (4096, 1028)]


def _large_host_call(seed, rng, api, binding, torch):
    if seed % 29 != 0:
        return 0
    duplex = seed % 499 == 0
    names = ["BC1", "BC3", "BPTC", "RGTC2", "ETC2_EAC", "BPTC_FLOAT", "EAC_R11"]
    fmt = F.BY_NAME[names[int(rng.integers(0, len(names)))]]
    if duplex:
        W, H = 8192, 4 * int(rng.integers(1024, 1100))
        if fmt.block_bytes == 8:
            H *= 2                                                     # 32+ MiB of blocks for the 8-byte formats too
    else:
        W, H = LARGE_GEOMETRIES[int(rng.integers(0, len(LARGE_GEOMETRIES)))]
    wb, hb = (W + 3) // 4, (H + 3) // 4
    data = ol.stream_u(fmt, wb * hb, seed=int(rng.integers(1, 1 << 40)))
    if rng.integers(0, 2) == 0 and fmt.name == "BPTC":
        data = data.copy().reshape(-1, 16); data[:, 0] |= 1; data = data.reshape(-1)      # all blocks valid: the result is then true
    dev = torch.from_numpy(np.ascontiguousarray(data)).cuda()
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    tiled = W % 4 == 0 and H % 4 == 0 and rng.integers(0, 3) == 0
    if tiled:
        want = binding.decompress_tiled_device(fmt, dev, wb, hb, status=status)
    else:
        want = binding.decompress_linear_device(fmt, dev, W, H, status=status)
    torch.cuda.synchronize()
    ok_h, got = api.tiled(fmt, data, wb, hb) if tiled else api.linear(fmt, data, W, H)
    where = (fmt.name, W, H, seed, "tiled" if tiled else "linear", "duplex" if duplex else "staged")
    assert np.array_equal(got, want.cpu().numpy()), ("large host call",) + where
    assert ok_h == bool(status.item() == 0), ("large host call ok",) + where
    return 1
