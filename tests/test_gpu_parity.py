"""GPU parity tests: the HIP kernels, called through the C ABI of libdetexhip.so, against
(1) the committed goldens produced by the compiled reference and (2) the CPU oracle on the same
seeded inputs.  Bit-exact everywhere: the whole path is integer / byte work.

Run on the GPU box:  python -m pytest tests -m gpu -x -q
Nothing here reads /root/reference.
"""
import hashlib
import os

import numpy as np
import pytest

import oracle_lib as ol
import streams
from detex_amd import formats as F

pytestmark = pytest.mark.gpu

sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
FMT_IDS = [f.name for f in F.FORMATS]


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def _dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a).reshape(-1)).cuda()


def _first_diff(a, b, per):
    a = np.asarray(a).reshape(-1, per); b = np.asarray(b).reshape(-1, per)
    bad = np.nonzero((a != b).any(axis=1))[0]
    return "first differing unit %d of %d differing: got %s want %s" % (bad[0], len(bad), a[bad[0]][:16], b[bad[0]][:16])


# ---- (i) the reference's bundled fixtures, end to end through the reference's own entry point ----
@pytest.mark.parametrize("fmt", [f for f in F.FORMATS if f.fixture], ids=lambda f: f.name)
def test_fixture_host_api(fmt, hiplib, golden_json, oracle):
    from detex_amd.ktx import read_ktx
    k = read_ktx(os.path.join(os.path.dirname(__file__), "golden", fmt.fixture))
    want = golden_json("fixtures.json")[fmt.name]
    for pf in F.accepted_pixel_formats(fmt):     # native, RGBX8<->RGBA8 and the in-kernel epilogue targets
        ok, out = hiplib.linear(fmt, k["data"], k["width"], k["height"], pixel_format=pf)
        g = want["0x%04X" % pf]
        assert ok == g["ok"]
        assert out.size == g["bytes"]
        assert sha(out) == g["sha256"], (fmt.name, hex(pf))
        ok_o, out_o = oracle.linear_to(fmt, k["data"], 64, 64, pf)
        assert np.array_equal(out, out_o)
        ok_t, out_t = hiplib.tiled(fmt, k["data"], 16, 16, pixel_format=pf)
        ok_ot, out_ot = oracle.tiled_to(fmt, k["data"], 16, 16, pf)
        assert ok_t == ok_ot and np.array_equal(out_t, out_ot), (fmt.name, hex(pf))


# ---- (ii) mode-forced vectors: every mode / invalid class, per-block results + ok flags ----------
@pytest.mark.parametrize("fmt", F.FORMATS, ids=FMT_IDS)
def test_forced_vectors_device(fmt, torch_cuda, forced_vectors):
    from detex_amd import binding
    torch = torch_cuda
    blocks = forced_vectors[fmt.name + "/in"]; want = forced_vectors[fmt.name + "/out"]; want_ok = forced_vectors[fmt.name + "/ok"]
    n = len(blocks)
    out, ok = binding.decompress_blocks_device(fmt, _dev(torch, blocks), n)
    torch.cuda.synchronize()
    out = out.cpu().numpy().reshape(n, -1); ok = ok.cpu().numpy()[:n]
    assert np.array_equal(ok, want_ok), "ok flags differ at %s" % np.nonzero(ok != want_ok)[0][:8]
    assert np.array_equal(out, want), _first_diff(out, want, want.shape[1])


@pytest.mark.parametrize("fmt", F.FORMATS, ids=FMT_IDS)
def test_mask_flag_matrix_device(fmt, torch_cuda, forced_vectors, golden_json, oracle):
    from detex_amd import binding
    torch = torch_cuda
    blocks = forced_vectors[fmt.name + "/in"]
    n = len(blocks)
    d_blocks = _dev(torch, blocks)
    golden = golden_json("maskflags.json")[fmt.name]
    for mask, flags in streams.MASK_FLAG_MATRIX:
        out, ok = binding.decompress_blocks_device(fmt, d_blocks, n, mode_mask=mask, flags=flags)
        torch.cuda.synchronize()
        out = out.cpu().numpy().reshape(n, -1); ok = ok.cpu().numpy()[:n]
        g = golden["%08X/%X" % (mask, flags)]
        want_ok = np.unpackbits(np.frombuffer(bytes.fromhex(g["ok_bits"]), np.uint8))[:n]
        assert np.array_equal(ok, want_ok), (fmt.name, hex(mask), flags)
        if sha(out) != g["sha256"]:
            ok_o, out_o = oracle.blocks(fmt, blocks, mask, flags)
            raise AssertionError("%s mask %08X flags %X: %s" % (fmt.name, mask, flags, _first_diff(out, out_o, out.shape[1])))


# ---- per-block host API (the 19 leaf functions + detexDecompressBlock) on the GPU ----------------
@pytest.mark.parametrize("fmt", F.FORMATS, ids=FMT_IDS)
def test_leaf_functions_host_api(fmt, hiplib, forced_vectors):
    blocks = forced_vectors[fmt.name + "/in"]; want = forced_vectors[fmt.name + "/out"]; want_ok = forced_vectors[fmt.name + "/ok"]
    step = max(1, len(blocks) // 48)
    for i in list(range(0, len(blocks), step)) + [len(blocks) - 1]:
        ok, out = hiplib.block(fmt, blocks[i])
        assert ok == bool(want_ok[i]), (fmt.name, i)
        if ok:
            assert np.array_equal(out, want[i]), (fmt.name, i)
    # generic entry: error text of the reference on a failing block (texture.c:63-64)
    bad = np.nonzero(want_ok == 0)[0]
    import ctypes
    out = np.zeros(16 * fmt.pixel_bytes, np.uint8)
    if len(bad):
        r = hiplib.lib.detexDecompressBlock(ol._ptr(blocks[bad[0]]), fmt.texture_format, 0xFFFFFFFF, 0, ol._ptr(out), F.native_pixel_format(fmt))
        assert not r
        assert hiplib.error() == "detexDecompressBlock: Decompress function for format 0x%08X returned error" % fmt.texture_format
    good = np.nonzero(want_ok == 1)[0][0]
    r = hiplib.lib.detexDecompressBlock(ol._ptr(blocks[good]), fmt.texture_format, 0xFFFFFFFF, 0, ol._ptr(out), F.native_pixel_format(fmt))
    assert r and np.array_equal(out, want[good])


# ---- the batched HOST-pointer block entry (detexhipDecompressBlocks): the loop over a leaf function as one call ------------------
@pytest.mark.parametrize("fmt", F.FORMATS, ids=FMT_IDS)
def test_batched_host_blocks_mask_flag_matrix(fmt, hiplib, forced_vectors, golden_json, oracle):
    """all forced vectors of the format (every mode and invalid class) through detexhipDecompressBlocks under every (mode_mask, flags)
    pair of the matrix: pixels (sha256) and per-block ok bits == the compiled reference's leaf function, the bool result == all(ok),
    the reference's error text on failure"""
    blocks = forced_vectors[fmt.name + "/in"]
    n = len(blocks)
    golden = golden_json("maskflags.json")[fmt.name]
    for mask, flags in streams.MASK_FLAG_MATRIX:
        r, ok, out = hiplib.blocks(fmt, blocks, mask, flags)
        g = golden["%08X/%X" % (mask, flags)]
        want_ok = np.unpackbits(np.frombuffer(bytes.fromhex(g["ok_bits"]), np.uint8))[:n]
        assert np.array_equal(ok, want_ok), (fmt.name, hex(mask), flags)
        assert r == bool(want_ok.all())
        if not r:
            assert hiplib.error() == "detexDecompressBlock: Decompress function for format 0x%08X returned error" % fmt.texture_format
        if sha(out) != g["sha256"]:
            ok_o, out_o = oracle.blocks(fmt, blocks, mask, flags)
            raise AssertionError("%s mask %08X flags %X: %s" % (fmt.name, mask, flags, _first_diff(out, out_o, out.shape[1])))


@pytest.mark.parametrize("name", ["BC1", "RGTC1", "BPTC", "BPTC_FLOAT", "ETC2_EAC", "EAC_SIGNED_R11"])
def test_batched_host_blocks_sizes(name, hiplib, oracle):
    """batch sizes on both sides of every internal boundary (one block: the leaf path; up to the pinned exchange's limit; staged through
    device buffers above it; ragged tails; a misaligned input pointer; no ok array): == the oracle's per-block decode"""
    fmt = F.BY_NAME[name]
    per = fmt.block_bytes + 16 * fmt.pixel_bytes + 1
    edge = (1280 << 10) // per
    for n in (1, 2, 63, 64, 65, 255, 256, 257, 1000, edge - 1, edge, edge + 1, edge + 300, 70001):
        data = ol.stream_u(fmt, n, seed=0xBA7C + n)
        want_ok, want = oracle.blocks(fmt, data)
        r, ok, out = hiplib.blocks(fmt, data)
        assert np.array_equal(ok.astype(bool), want_ok), (name, n)
        assert np.array_equal(out, want), (name, n, _first_diff(out, want, want.shape[1]))
        assert r == bool(want_ok.all())
    # unaligned input, no ok array
    n = 777
    raw = np.zeros(n * fmt.block_bytes + 1, np.uint8)
    raw[1:] = ol.stream_u(fmt, n, seed=0x0DD)
    want_ok, want = oracle.blocks(fmt, raw[1:])
    r, ok, out = hiplib.blocks(fmt, raw[1:], want_ok=False)
    assert ok is None and np.array_equal(out, want) and r == bool(want_ok.all())
    # zero blocks: true, nothing touched; an unknown format: false + message
    f = hiplib.lib.detexhipDecompressBlocks
    assert f(fmt.texture_format, None, 0, 0xFFFFFFFF, 0, None, None)
    assert not f(0x14800334, ol._ptr(raw), 1, 0xFFFFFFFF, 0, ol._ptr(raw), None) and "not a block-compressed format" in hiplib.error()


# ---- (iii) clipped sizes through the host API and the device API with a padded pitch -------------
@pytest.mark.parametrize("fmt", F.FORMATS, ids=FMT_IDS)
def test_clipped_sizes(fmt, hiplib, torch_cuda, forced_vectors, clip_vectors):
    from detex_amd import binding
    torch = torch_cuda
    flat = forced_vectors[fmt.name + "/in"].reshape(-1)
    px = fmt.pixel_bytes
    for (w, h) in streams.CLIP_SIZES:
        wb, hb = (w + 3) // 4, (h + 3) // 4
        data = np.resize(flat, wb * hb * fmt.block_bytes)
        want = clip_vectors["%s/%dx%d" % (fmt.name, w, h)]
        want_ok = bool(clip_vectors["%s/%dx%d/ok" % (fmt.name, w, h)][0])
        ok, out = hiplib.linear(fmt, data, w, h)
        assert ok == want_ok, (fmt.name, w, h)
        assert np.array_equal(out, want), (fmt.name, w, h, _first_diff(out, want, px))
        # device tier: pitch padded to 64 bytes, canary bytes beyond each row must survive
        pitch = ((w * px + 63) // 64) * 64 + 64
        canvas = torch.full((h * pitch,), 0xA5, dtype=torch.uint8, device="cuda")
        status = torch.zeros(1, dtype=torch.int32, device="cuda")
        binding.decompress_linear_device(fmt, _dev(torch, data), w, h, out=canvas, pitch=pitch, status=status)
        torch.cuda.synchronize()
        got = canvas.cpu().numpy().reshape(h, pitch)
        assert np.array_equal(got[:, :w * px].reshape(-1), want), (fmt.name, w, h, "device pitch")
        assert (got[:, w * px:] == 0xA5).all(), (fmt.name, w, h, "wrote outside the image")
        assert bool(status.item() == 0) == want_ok
        if (w, h) in streams.CLIP_SIZES_CONVERTED:      # the same clipped geometry into the epilogue targets
            for pf in F.accepted_pixel_formats(fmt):
                if not F.epilogue_kind(fmt, pf):
                    continue
                want_c = clip_vectors["%s/%dx%d/pf%04X" % (fmt.name, w, h, pf)]
                ok, out = hiplib.linear(fmt, data, w, h, pixel_format=pf)
                assert ok == want_ok and np.array_equal(out, want_c), (fmt.name, w, h, hex(pf))
                tpx = 1 + ((pf & 0xF00) >> 8)
                pitch = w * tpx + 7                       # odd pitch: 24-bit pixels are byte-addressed
                if tpx != 3:
                    pitch = ((w * tpx + 15) // 16) * 16 + 16
                canvas = torch.full((h * pitch,), 0xA5, dtype=torch.uint8, device="cuda")
                binding.decompress_linear_device(fmt, _dev(torch, data), w, h, out=canvas, pitch=pitch, pixel_format=pf)
                torch.cuda.synchronize()
                got = canvas.cpu().numpy().reshape(h, pitch)
                assert np.array_equal(got[:, :w * tpx].reshape(-1), want_c), (fmt.name, w, h, hex(pf), "device pitch")
                assert (got[:, w * tpx:] == 0xA5).all()


# ---- random streams vs the oracle at a size the oracle finishes in well under a second ----------
@pytest.mark.parametrize("fmt", F.FORMATS, ids=FMT_IDS)
def test_random_stream_vs_oracle(fmt, torch_cuda, oracle):
    from detex_amd import binding
    torch = torch_cuda
    W, H = 1024, 512
    data = ol.stream_u(fmt, (W // 4) * (H // 4), seed=0xC0FFEE + fmt.index)
    ok_o, want = oracle.linear(fmt, data, W, H)
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    out = binding.decompress_linear_device(fmt, _dev(torch, data), W, H, status=status)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.array_equal(got, want), _first_diff(got, want, 16 * fmt.pixel_bytes)
    assert bool(status.item() == 0) == ok_o
    # tiled layout of the same stream
    ok_t, want_t = oracle.tiled(fmt, data, W // 4, H // 4)
    got_t = binding.decompress_tiled_device(fmt, _dev(torch, data), W // 4, H // 4)
    torch.cuda.synchronize()
    assert np.array_equal(got_t.cpu().numpy(), want_t)


NARROW = [f for f in F.FORMATS if f.pixel_bytes <= 2]


@pytest.mark.parametrize("fmt", NARROW, ids=lambda f: f.name)
def test_narrow_pixel_grouped_rows(fmt, torch_cuda, oracle):
    """1- and 2-byte pixels take the several-blocks-per-lane kernel when width_in_blocks % G == 0 and rows are 16-byte
    aligned, the one-block-per-lane kernel otherwise: block-row widths on both sides of the condition, group counts that
    do not fill the last workgroup, padded pitches (aligned and not), a canary after every row."""
    from detex_amd import binding
    torch = torch_cuda
    px = fmt.pixel_bytes
    for wb, hb, pad in [(4, 3, 0), (6, 5, 0), (7, 2, 0), (20, 9, 16), (257, 3, 0), (258, 3, 8), (260, 5, 32), (1028, 2, 4), (64, 33, 0)]:
        W, H = wb * 4, hb * 4
        data = ol.stream_u(fmt, wb * hb, seed=0x6A0 + 131 * wb + fmt.index)
        ok_o, want = oracle.linear(fmt, data, W, H)
        pitch = W * px + pad
        out = torch.full((H * pitch,), 0xA5, dtype=torch.uint8, device="cuda")
        status = torch.zeros(1, dtype=torch.int32, device="cuda")
        binding.decompress_linear_device(fmt, _dev(torch, data), W, H, out=out, pitch=pitch, status=status)
        torch.cuda.synchronize()
        got = out.cpu().numpy().reshape(H, pitch)
        assert np.array_equal(got[:, :W * px].reshape(-1), want.reshape(-1)), (fmt.name, wb, hb, pad)
        assert (got[:, W * px:] == 0xA5).all(), (fmt.name, wb, hb, pad)
        assert bool(status.item() == 0) == ok_o


def _etc2_is_planar(colour_bytes):
    """ETC2 mode selection (decompress-etc.c:324-366) on the 8 colour bytes of each block: differential bit set,
    R and G sums in range, B sum out of range."""
    b = colour_bytes.astype(np.int32)

    def total(x):
        d = x & 7
        return (x >> 3) + np.where(d & 4, d - 8, d)
    r, g, bl = total(b[:, 0]), total(b[:, 1]), total(b[:, 2])
    inr = lambda v: (v >= 0) & (v <= 31)
    return ((b[:, 3] & 2) != 0) & inr(r) & inr(g) & ~inr(bl)


@pytest.mark.parametrize("name", ["ETC2", "ETC2_PUNCHTHROUGH", "ETC2_EAC"])
def test_etc2_planar_blocks_per_wave(name, torch_cuda, oracle):
    """A wave with up to eight planar blocks decodes them cooperatively (one texel per lane through LDS), a wave with more
    decodes them in their own lanes: streams with 0..10, 63 and 64 planar blocks per group of 64, at the start, the end and
    scattered positions, linear and block-major layouts, ragged block counts, and the per-block API with the planar mode
    masked out (those blocks fail before the cooperative part)."""
    from detex_amd import binding
    torch = torch_cuda
    fmt = F.BY_NAME[name]
    bb = fmt.block_bytes
    pool = ol.stream_u(fmt, 1 << 16, seed=0x91A7A + fmt.index).reshape(-1, bb)
    planar = _etc2_is_planar(pool[:, bb - 8:])
    if name == "ETC2":                                            # the classifier above against the restated detexGetModeETC2
        assert np.array_equal(planar, oracle.modes(fmt, pool.reshape(-1)) == 4)
    p_pool, o_pool = pool[planar], pool[~planar]
    assert len(p_pool) > 1000 and len(o_pool) > 20000
    rng = np.random.default_rng(0x5EED + fmt.index)
    counts = [0, 1, 2, 3, 4, 5, 7, 8, 9, 10, 63, 64, 1, 8, 4, 0]
    waves = []
    for w, k in enumerate(counts):
        blk = o_pool[rng.integers(0, len(o_pool), 64)].copy()
        where = np.arange(k) if w % 3 == 0 else (63 - np.arange(k) if w % 3 == 1 else rng.choice(64, k, replace=False))
        blk[where] = p_pool[rng.integers(0, len(p_pool), k)]
        waves.append(blk)
    data = np.concatenate(waves).reshape(-1)                      # 1024 blocks: 64 x 16 blocks
    W, H = 256, 64
    ok_o, want = oracle.linear(fmt, data, W, H)
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    got = binding.decompress_linear_device(fmt, _dev(torch, data), W, H, status=status)
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), want)
    assert bool(status.item() == 0) == ok_o
    ok_t, want_t = oracle.tiled(fmt, data, W // 4, H // 4)
    got_t = binding.decompress_tiled_device(fmt, _dev(torch, data), W // 4, H // 4)
    torch.cuda.synchronize()
    assert np.array_equal(got_t.cpu().numpy(), want_t)
    # ragged counts (the last wave is partly idle) and a mode mask without the planar mode
    for n in (1, 37, 64, 65, 700, 1023):
        for mask in (0xFFFFFFFF, 0x0F, 0x10):
            ok_b, want_b = oracle.blocks(fmt, data[:n * bb], mode_mask=mask)
            got_b, got_ok = binding.decompress_blocks_device(fmt, _dev(torch, data[:n * bb]), n, mode_mask=mask)
            torch.cuda.synchronize()
            assert np.array_equal(got_ok.cpu().numpy()[:n].astype(bool), ok_b), (n, mask)
            assert np.array_equal(got_b.cpu().numpy().reshape(-1)[:n * 16 * fmt.pixel_bytes], want_b.reshape(-1)), (n, mask)


@pytest.mark.parametrize("name,pf", [(f.name, pf) for f in F.FORMATS for pf in F.accepted_pixel_formats(f) if F.epilogue_kind(f, pf)])
def test_epilogue_targets_random_stream(name, pf, torch_cuda, oracle):
    """in-kernel pixel-format epilogues (BGRA8/BGRX8/RGB8, FLOAT_BGRX16) vs oracle decode + convert"""
    from detex_amd import binding
    torch = torch_cuda
    fmt = F.BY_NAME[name]
    W, H = 512, 256
    data = ol.stream_u(fmt, (W // 4) * (H // 4), seed=0xE91 + fmt.index)
    ok_o, want = oracle.linear_to(fmt, data, W, H, pf)
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    out = binding.decompress_linear_device(fmt, _dev(torch, data), W, H, pixel_format=pf, status=status)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), want)
    assert bool(status.item() == 0) == ok_o
    ok_t, want_t = oracle.tiled_to(fmt, data, W // 4, H // 4, pf)
    got_t = binding.decompress_tiled_device(fmt, _dev(torch, data), W // 4, H // 4, pixel_format=pf)
    torch.cuda.synchronize()
    assert np.array_equal(got_t.cpu().numpy(), want_t)


def test_full_size_digest_epilogue_targets(torch_cuda, golden_json):
    from detex_amd import binding
    torch = torch_cuda
    dg = golden_json("digests_8192.json")
    W, H = dg["width"], dg["height"]
    for key, g in dg["streams"].items():
        if "/pf" not in key:
            continue
        name, _, pfs = key.split("/")
        fmt, pf = F.BY_NAME[name], int(pfs[2:], 16)
        data = ol.stream_u(fmt, (W // 4) * (H // 4))
        assert sha(data) == g["in_sha256"]
        status = torch.zeros(1, dtype=torch.int32, device="cuda")
        out = binding.decompress_linear_device(fmt, _dev(torch, data), W, H, pixel_format=pf, status=status)
        torch.cuda.synchronize()
        assert out.numel() == g["bytes"] and bool(status.item() == 0) == g["ok"]
        assert sha(out.cpu().numpy()) == g["sha256"], key
        del out
        torch.cuda.empty_cache()


@pytest.mark.parametrize("fmt", F.FORMATS, ids=FMT_IDS)
def test_blocks_ragged_counts(fmt, torch_cuda, oracle):
    """block-major kernels with block counts that end inside a wave / a workgroup: the staged row stores
    must write exactly n*16*px bytes (canary behind the end) and the right blocks"""
    from detex_amd import binding
    torch = torch_cuda
    px = fmt.pixel_bytes
    for n in (1, 3, 63, 65, 255, 257, 1000 + 37):
        data = ol.stream_u(fmt, n, seed=0xBEEF + 7 * n + fmt.index)
        ok_o, want = oracle.blocks(fmt, data.reshape(-1, fmt.block_bytes))
        canvas = torch.full((n * 16 * px + 256,), 0xA5, dtype=torch.uint8, device="cuda")
        out, ok = binding.decompress_blocks_device(fmt, _dev(torch, data), n, out=canvas)
        torch.cuda.synchronize()
        got = canvas.cpu().numpy()
        assert np.array_equal(got[:n * 16 * px], want.reshape(-1)), (fmt.name, n)
        assert (got[n * 16 * px:] == 0xA5).all(), (fmt.name, n, "wrote past the end")
        assert np.array_equal(ok.cpu().numpy()[:n].astype(bool), ok_o), (fmt.name, n)
        # the texture-driver form (no per-block flags) of the same count
        _, want_t = oracle.tiled(fmt, data, n, 1)
        canvas.fill_(0xA5)
        binding.decompress_tiled_device(fmt, _dev(torch, data), n, 1, out=canvas)
        torch.cuda.synchronize()
        got = canvas.cpu().numpy()
        assert np.array_equal(got[:n * 16 * px], want_t.reshape(-1)) and (got[n * 16 * px:] == 0xA5).all(), (fmt.name, n, "tiled")


# ---- (iv) BASELINE.json's full size: 8192x8192 streams against the reference's digests ----------
HEADLINE = ["BC1", "BC3", "BPTC", "ETC2", "ETC2_EAC", "BPTC_FLOAT"]


@pytest.mark.parametrize("name", [f.name for f in F.FORMATS])
def test_full_size_digest(name, torch_cuda, golden_json):
    from detex_amd import binding
    torch = torch_cuda
    fmt = F.BY_NAME[name]
    dg = golden_json("digests_8192.json")
    W, H = dg["width"], dg["height"]
    kinds = ["U"] + (["M"] if name in ("BPTC", "BPTC_FLOAT", "BPTC_SIGNED_FLOAT") else []) + (["C"] if fmt.fixture else [])
    for kind in kinds:                              # U uniform random, M modes equiprobable, C the bundled fixture tiled
        g = dg["streams"]["%s/%s" % (name, kind)]   # native target; epilogue targets: test_full_size_digest_epilogue_targets
        data = streams.make_stream(kind, fmt, W // 4, H // 4)
        assert sha(data) == g["in_sha256"], "stream generator drifted"
        status = torch.zeros(1, dtype=torch.int32, device="cuda")
        out = binding.decompress_linear_device(fmt, _dev(torch, data), W, H, status=status)
        torch.cuda.synchronize()
        assert bool(status.item() == 0) == g["ok"], (name, kind)
        assert sha(out.cpu().numpy()) == g["sha256"], (name, kind)
        del out
        torch.cuda.empty_cache()


def test_seeded_fuzz_slice(torch_cuda, oracle):
    """A fixed slice of the randomised sweep of tools/gpu_fuzz.py (tests/fuzz.py): 300 seeds x 19 formats, ~20 k decode calls
    -- non-power-of-two block widths, all pixel sizes, padded pitches, epilogue targets, block-major, random mode masks."""
    from detex_amd import binding
    import fuzz
    binding.load()
    cases = sum(fuzz.run_seed(seed, oracle, binding, torch_cuda) for seed in range(1000, 1300))
    assert cases >= 300 * 19 * 3


@pytest.mark.parametrize("name", ["BPTC_FLOAT", "BPTC_SIGNED_FLOAT", "BC3", "BPTC", "RGTC2", "RGTC1", "SIGNED_RGTC1", "EAC_RG11"])
def test_wide_non_power_of_two_band_vs_oracle(name, torch_cuda, oracle):
    """8000 pixels wide (2000 blocks per row: the division branch of split_index, waves and 64-bit-pixel transposes that
    straddle block rows, several-blocks-per-lane rows) on a band of 64 block rows, against the ORACLE (test_properties_full_size
    compares such widths with themselves only)."""
    from detex_amd import binding
    torch = torch_cuda
    fmt = F.BY_NAME[name]
    W, H = 8000, 256
    data = ol.stream_u(fmt, (W // 4) * (H // 4), seed=0x8000 + fmt.index)
    ok_o, want = oracle.linear(fmt, data, W, H)
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    got = binding.decompress_linear_device(fmt, _dev(torch, data), W, H, status=status)
    torch.cuda.synchronize()
    g = got.cpu().numpy().reshape(-1)
    assert np.array_equal(g, want.reshape(-1)), (name, _first_diff(g, want.reshape(-1), fmt.pixel_bytes))
    assert bool(status.item() == 0) == ok_o
    _, want_t = oracle.tiled(fmt, data, W // 4, H // 4)
    got_t = binding.decompress_tiled_device(fmt, _dev(torch, data), W // 4, H // 4)
    torch.cuda.synchronize()
    assert np.array_equal(got_t.cpu().numpy(), want_t), name


@pytest.mark.parametrize("name", ["BC1", "BPTC", "BPTC_FLOAT", "RGTC2", "EAC_R11"])
def test_block_grid_smaller_or_larger_than_the_image(name, torch_cuda, oracle):
    """texture.c:116-136 with a block grid that does not match the image: a grid that covers only part of the image leaves the
    other pixels untouched (canary), a grid that is larger has its surplus blocks dropped -- through the staged kernel (dword-aligned
    rows), for widths on both sides of a 256-block tile"""
    from detex_amd import binding
    torch = torch_cuda
    fmt = F.BY_NAME[name]
    px = fmt.pixel_bytes
    for (W, H, wb, hb) in ((1000, 52, 200, 10), (1000, 52, 260, 15), (2052, 20, 513, 5), (2052, 20, 300, 7), (64, 64, 20, 3), (36, 12, 9, 5)):
        data = ol.stream_u(fmt, wb * hb, seed=0x6A1D + wb)
        want = np.full(W * H * px, 0xA5, np.uint8)
        ok_o = oracle.lib.orc_decompress_linear(fmt.index, ol._ptr(data), W, H, wb, hb, ol._ptr(want))
        canvas = torch.full((W * H * px + 256,), 0xA5, dtype=torch.uint8, device="cuda")
        status = torch.zeros(1, dtype=torch.int32, device="cuda")
        binding.decompress_linear_device(fmt, _dev(torch, data), W, H, out=canvas, status=status, width_in_blocks=wb, height_in_blocks=hb)
        torch.cuda.synchronize()
        got = canvas.cpu().numpy()
        assert np.array_equal(got[:W * H * px], want), (name, W, H, wb, hb, _first_diff(got[:W * H * px], want, px))
        assert (got[W * H * px:] == 0xA5).all() and bool(status.item() == 0) == bool(ok_o), (name, W, H, wb, hb)


def test_clipped_large_digests(torch_cuda, golden_json, oracle):
    """Large textures whose width / height are not multiples of four (texture.c:116-120, 132-136) against digests of the
    compiled reference's output: the interior goes through the throughput kernel (rows 16-byte aligned, or only dword-aligned
    when an odd width puts every other row 4 / 8 / 12 bytes off), the last block column / row through the per-pixel kernel,
    and widths whose rows are not even dword-aligned (R8 4093 wide, RGB8 4094 wide) through that kernel as a whole.  A canary
    behind the image must survive; the first 64 rows are also compared with the oracle so a failure names a pixel."""
    from detex_amd import binding
    torch = torch_cuda
    dg = golden_json("digests_8192.json")["clipped"]
    for key, g in dg.items():
        parts = key.split("/")
        fmt = F.BY_NAME[parts[0]]
        W, H = (int(v) for v in parts[1].split("x"))
        pf = int(parts[2][2:], 16) if len(parts) > 2 else F.native_pixel_format(fmt)
        px = 1 + ((pf & 0xF00) >> 8)
        wb, hb = (W + 3) // 4, (H + 3) // 4
        data = ol.stream_u(fmt, wb * hb)
        assert sha(data) == g["in_sha256"]
        canvas = torch.full((W * H * px + 4096,), 0xA5, dtype=torch.uint8, device="cuda")
        status = torch.zeros(1, dtype=torch.int32, device="cuda")
        binding.decompress_linear_device(fmt, _dev(torch, data), W, H, out=canvas, pixel_format=pf, status=status)
        torch.cuda.synchronize()
        got = canvas.cpu().numpy()
        assert (got[W * H * px:] == 0xA5).all(), (key, "wrote past the image")
        rows = min(64, H)
        _, want = oracle.linear_to(fmt, data[:((rows + 3) // 4) * wb * fmt.block_bytes], W, rows, pf)
        assert np.array_equal(got[:rows * W * px], want), (key, _first_diff(got[:rows * W * px], want, px))
        assert bool(status.item() == 0) == g["ok"], key
        assert g["bytes"] == W * H * px and sha(got[:W * H * px]) == g["sha256"], key
        del canvas
        torch.cuda.empty_cache()


# ---- size-independent properties at full size -----------------------------------------------------
@pytest.mark.parametrize("name,W,H", [("BC1", 4096, 4096), ("BPTC", 4096, 4096), ("BPTC_FLOAT", 4096, 4096), ("BPTC_SIGNED_FLOAT", 4096, 2048),
                                      ("RGTC1", 4096, 4096), ("EAC_R11", 4096, 2048), ("BPTC_FLOAT", 8000, 2000), ("BC3", 8000, 2000),
                                      ("RGTC2", 8000, 1000)])
def test_properties_full_size(name, W, H, torch_cuda):
    """(a) row-band sharding: decoding block-row bands separately into the same image is identical to one whole-image
    decode (the multi-GPU decomposition of SURVEY 8e) -- for 1-, 2-, 4- and 8-byte pixels (the 64-bit ones leave through
    the per-wave LDS transpose) and for widths that are not powers of two (waves straddle block rows);
    (b) linear and tiled layouts hold the same texels; (c) decode is idempotent."""
    from detex_amd import binding
    torch = torch_cuda
    fmt = F.BY_NAME[name]
    wb, hb = W // 4, H // 4
    d_blocks = _dev(torch, ol.stream_u(fmt, wb * hb, seed=99))
    whole = binding.decompress_linear_device(fmt, d_blocks, W, H)
    again = binding.decompress_linear_device(fmt, d_blocks, W, H)
    banded = torch.zeros_like(whole)
    px = fmt.pixel_bytes
    for g in range(8):
        r0, r1 = g * hb // 8, (g + 1) * hb // 8
        binding.decompress_linear_device(fmt, d_blocks[r0 * wb * fmt.block_bytes:], W, (r1 - r0) * 4,
                                         out=banded[r0 * 4 * W * px:])
    tiled = binding.decompress_tiled_device(fmt, d_blocks, wb, hb)
    torch.cuda.synchronize()
    assert torch.equal(whole, again)
    assert torch.equal(whole, banded)
    t = tiled.view(hb, wb, 4, 4 * px).permute(0, 2, 1, 3).reshape(-1)
    assert torch.equal(t, whole)


def test_unsupported_targets_fail_loudly(hiplib):
    fmt = F.BY_NAME["BC1"]
    data = np.zeros(16 * 16 * 8, np.uint8)
    out = np.full(64 * 64 * 8, 0x5A, np.uint8)
    ok, out = hiplib.linear(fmt, data, 64, 64, pixel_format=0x6735, out=out)    # FLOAT_RGBA16: unreachable in the reference too
    assert not ok and not out.any()
    assert "outside the block-decode path" in hiplib.error()


def test_signed_bc6h_extreme_magnitudes(torch_cuda, oracle):
    """interpolated value -32768 -> half 0xFC00 (see tests/test_host_logic.py for the fixture)"""
    from detex_amd import binding
    torch = torch_cuda
    fmt = F.BY_NAME["BPTC_SIGNED_FLOAT"]
    blocks = np.load(os.path.join(os.path.dirname(__file__), "golden", "bc6h_signed_extreme_blocks.npy"))
    ok_o, want = oracle.blocks(fmt, blocks)
    out, ok = binding.decompress_blocks_device(fmt, _dev(torch, blocks), len(blocks))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().reshape(len(blocks), -1), want)
    data = np.resize(blocks.reshape(-1), (256 // 4) * (64 // 4) * 16)      # the same blocks through the linear kernel
    _, want_l = oracle.linear(fmt, data, 256, 64)
    got = binding.decompress_linear_device(fmt, _dev(torch, data), 256, 64)
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), want_l)


@pytest.mark.parametrize("name,world,g", [("BC1", 4, g) for g in range(4)] + [("BPTC_FLOAT", 8, g) for g in range(8)] + [("BC1", 8, 5)])
def test_sharded_config_bands_whole_digest(name, world, g, torch_cuda, golden_json):
    """EVERY band of the sharded 32768^2 configurations (BASELINE configs[4] / north_star: BC1 over 4 GPUs, BC6H over 8;
    texture.c:105-145 on those inputs): rank g's launch exactly as bench.py --gpus N issues it -- the band's blocks from the
    GLOBAL stream offset (sharding.shard_of -> stream_u_slice), one detexhipDecompressTextureLinearDevice call over the band --
    and the sha256 of the WHOLE band (1 GiB of pixels) against the compiled reference's (tools/make_goldens.py bands_all)"""
    from detex_amd import binding, sharding
    torch = torch_cuda
    fmt = F.BY_NAME[name]
    side = 32768
    gold = golden_json("digests_8192.json")["bands_all"]["%s/%d/%dof%d" % (name, side, g, world)]
    sh = sharding.shard_of(g, world, fmt, side, side)
    assert (sh.row0, sh.row1) == (gold["row0"], gold["row1"])
    data = ol.stream_u_slice(fmt, sh.in_offset // fmt.block_bytes, sh.in_bytes // fmt.block_bytes)
    assert sha(data) == gold["in_sha256"]
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    out = binding.decompress_linear_device(fmt, _dev(torch, data), side, sh.px_rows, status=status)
    torch.cuda.synchronize()
    assert out.numel() == gold["bytes"] == sh.out_bytes and bool(status.item() == 0) == gold["ok"]
    assert sha(out.cpu().numpy()) == gold["sha256"]
    del out
    torch.cuda.empty_cache()


def test_maximum_size_texture(torch_cuda, oracle):
    """the largest single texture the API takes (32768 x 32768 BC1: 4 GiB of pixels in one launch): 64-bit addressing and
    the block index arithmetic, checked bit-exact at the top, middle and bottom block rows (whole bands of this width
    are digested in test_sharded_config_bands_whole_digest)"""
    from detex_amd import binding
    torch = torch_cuda
    name, W, H = "BC1", 32768, 32768
    fmt = F.BY_NAME[name]
    wb, hb = W // 4, H // 4
    data = np.random.default_rng(5).integers(0, 256, size=wb * hb * fmt.block_bytes, dtype=np.uint8)
    d = torch.from_numpy(data).cuda()
    out = torch.empty(W * H * fmt.pixel_bytes, dtype=torch.uint8, device="cuda")
    binding.decompress_linear_device(fmt, d, W, H, out=out)
    torch.cuda.synchronize()
    rows = 4
    for r0 in (0, hb // 2 - 1, hb - rows):
        _, want = oracle.linear(fmt, data[r0 * wb * fmt.block_bytes:(r0 + rows) * wb * fmt.block_bytes], W, rows * 4)
        got = out[r0 * 4 * W * fmt.pixel_bytes:(r0 + rows) * 4 * W * fmt.pixel_bytes].cpu().numpy()
        assert np.array_equal(got, want), (name, r0)
    del d, out
    torch.cuda.empty_cache()


@pytest.mark.parametrize("name", ["BC1", "BPTC_FLOAT"])
def test_whole_32768_image_in_one_call_banded_read_ahead(name, torch_cuda, golden_json):
    """the WHOLE 32768^2 image of the sharded configurations through ONE detexhipDecompressTextureLinearDevice call.  Its blocks (512 MiB /
    1 GiB) exceed the 256 MiB Infinity Cache, so the entry decodes it in bands of <= 128 MiB of blocks, each read into that cache by a
    read-only pass first (device_tier.cpp; detexhipSetReadAhead, on by default): every one of the eight eighths of the image digests to the
    compiled reference's sha256 (tools/make_goldens.py bands_all), the status word is the reference's result, and the same call with the
    read-ahead switched off (one launch) writes the identical image"""
    from detex_amd import binding, sharding
    torch = torch_cuda
    fmt = F.BY_NAME[name]
    side = 32768
    gold = golden_json("digests_8192.json")["bands_all"]
    whole = sharding.shard_of(0, 1, fmt, side, side)
    data = ol.stream_u_slice(fmt, 0, whole.in_bytes // fmt.block_bytes)
    d = _dev(torch, data)
    del data
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    assert binding.set_read_ahead(True) in (0, 1)
    try:
        out = binding.decompress_linear_device(fmt, d, side, side, status=status)
        torch.cuda.synchronize()
        all_ok = True
        for e in range(8):
            g = gold["%s/%d/%dof8" % (name, side, e)]
            piece = out[e * g["bytes"]:(e + 1) * g["bytes"]]
            assert sha(piece.cpu().numpy()) == g["sha256"], (name, "eighth", e)
            all_ok = all_ok and g["ok"]
        assert bool(status.item() == 0) == all_ok
        binding.set_read_ahead(False)
        status2 = torch.zeros(1, dtype=torch.int32, device="cuda")
        out2 = binding.decompress_linear_device(fmt, d, side, side, status=status2)
        torch.cuda.synchronize()
        assert torch.equal(out, out2) and status.item() == status2.item()
    finally:
        binding.set_read_ahead(True)
    del d, out, out2
    torch.cuda.empty_cache()


@pytest.mark.parametrize("layout", ["linear", "tiled"])
def test_read_ahead_mode_2_bands_and_matches_one_launch(layout, torch_cuda, oracle):
    """detexhipSetReadAhead(2): every texture with >= 1 MiB of blocks goes in bands of <= 128 MiB of blocks behind a read-only pass -- here
    BPTC 16384^2 (256 MiB of blocks: two bands) and a 2 MiB one (one band), both layouts: identical bytes and status word to the single
    launch of mode 0, and the first block rows equal to the oracle's"""
    from detex_amd import binding
    torch = torch_cuda
    fmt = F.BY_NAME["BPTC"]
    for side in (16384, 1448 // 4 * 4):
        wb = hb = side // 4
        data = ol.stream_u(fmt, wb * hb, seed=0x2EAD + side)
        d = _dev(torch, data)
        outs, stats = [], []
        try:
            for mode in (0, 2):
                binding.set_read_ahead(mode)
                status = torch.zeros(1, dtype=torch.int32, device="cuda")
                if layout == "tiled":
                    out = binding.decompress_tiled_device(fmt, d, wb, hb, status=status)
                else:
                    out = binding.decompress_linear_device(fmt, d, side, side, status=status)
                torch.cuda.synchronize()
                outs.append(out); stats.append(int(status.item()))
        finally:
            binding.set_read_ahead(1)
        assert torch.equal(outs[0], outs[1]) and stats[0] == stats[1], (layout, side)
        rows = 4
        sub = data[:rows * wb * fmt.block_bytes]
        if layout == "tiled":
            _, want = oracle.tiled_to(fmt, sub, wb, rows, F.native_pixel_format(fmt))
        else:
            _, want = oracle.linear(fmt, sub, side, rows * 4)
        assert np.array_equal(outs[1][:want.size].cpu().numpy(), want.reshape(-1)), (layout, side)
        del d, outs
        torch.cuda.empty_cache()


def test_empty_inputs(hiplib, torch_cuda):
    """empty textures / zero blocks: the reference's loops simply do not run (texture.c:111-144 -> true, nothing
    written); no launch with an empty grid, no error text"""
    from detex_amd import binding
    torch = torch_cuda
    for name in ("BC1", "BPTC", "BPTC_FLOAT", "RGTC1"):
        fmt = F.BY_NAME[name]
        for (w, h) in ((0, 0), (0, 8), (8, 0)):
            out = np.full(64, 0xA5, np.uint8)
            tex = hiplib._texture(fmt, np.zeros(16, np.uint8), w, h)
            import ctypes
            assert hiplib.lib.detexDecompressTextureLinear(ctypes.byref(tex), ol._ptr(out), F.native_pixel_format(fmt))
            assert hiplib.lib.detexDecompressTextureTiled(ctypes.byref(tex), ol._ptr(out), F.native_pixel_format(fmt))
            assert (out == 0xA5).all()
        canvas = torch.full((256,), 0xA5, dtype=torch.uint8, device="cuda")
        status = torch.zeros(1, dtype=torch.int32, device="cuda")
        binding.decompress_linear_device(fmt, canvas[:16], 0, 0, out=canvas, status=status)
        binding.decompress_tiled_device(fmt, canvas[:16], 0, 0, out=canvas, status=status)
        binding.decompress_blocks_device(fmt, canvas[:16], 0, out=canvas)
        torch.cuda.synchronize()
        assert (canvas.cpu().numpy() == 0xA5).all() and status.item() == 0


def test_device_tier_is_graph_capturable(torch_cuda, oracle):
    """a device-tier call is one asynchronous launch with no allocation or synchronisation inside, so it
    can be captured into a hipGraph: capture three decodes (linear, tiled, mip levels via separate calls), replay twice
    on fresh inputs and compare"""
    from detex_amd import binding
    torch = torch_cuda
    fa, fb = F.BY_NAME["BC1"], F.BY_NAME["BPTC"]
    W, H = 512, 256
    n = (W // 4) * (H // 4)
    in_a = torch.zeros(n * fa.block_bytes, dtype=torch.uint8, device="cuda")
    in_b = torch.zeros(n * fb.block_bytes, dtype=torch.uint8, device="cuda")
    out_a = torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda")
    out_b = torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda")
    out_t = torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda")
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    binding.decompress_linear_device(fa, in_a, W, H, out=out_a)      # warm-up outside the capture (module load)
    binding.decompress_linear_device(fb, in_b, W, H, out=out_b, status=status)
    binding.decompress_tiled_device(fb, in_b, W // 4, H // 4, out=out_t)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        binding.decompress_linear_device(fa, in_a, W, H, out=out_a)
        binding.decompress_linear_device(fb, in_b, W, H, out=out_b, status=status)
        binding.decompress_tiled_device(fb, in_b, W // 4, H // 4, out=out_t)
    for rep in range(2):
        da = ol.stream_u(fa, n, seed=0x6A0 + rep); db = ol.stream_u(fb, n, seed=0x6B0 + rep)
        in_a.copy_(torch.from_numpy(da)); in_b.copy_(torch.from_numpy(db))
        status.zero_(); out_a.zero_(); out_b.zero_(); out_t.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(out_a.cpu().numpy(), oracle.linear(fa, da, W, H)[1]), rep
        ok_b, want_b = oracle.linear(fb, db, W, H)
        assert np.array_equal(out_b.cpu().numpy(), want_b) and bool(status.item() == 0) == ok_b, rep
        assert np.array_equal(out_t.cpu().numpy(), oracle.tiled(fb, db, W // 4, H // 4)[1]), rep


# ---- re-entrancy of the host-pointer tier (the reference is re-entrant; per-thread stream + staging here) ----
def test_host_api_concurrent_threads(hiplib, oracle):
    """eight host threads decode different formats / sizes through detexDecompressTextureLinear and the leaf
    functions at the same time (ctypes drops the GIL); every result and every thread-local error text is its own"""
    import threading
    names = ["BC1", "BPTC", "ETC2_EAC", "BPTC_FLOAT", "RGTC2", "EAC_SIGNED_R11", "BC3", "SIGNED_RGTC1"]
    jobs = []
    for k, name in enumerate(names):
        fmt = F.BY_NAME[name]
        w, h = 256 + 64 * k, 128 + 36 * (k % 3)           # some sizes are not multiples of 4 blocks wide
        data = ol.stream_u(fmt, ((w + 3) // 4) * ((h + 3) // 4), seed=0x7EAD + k)
        jobs.append((fmt, w, h, data, oracle.linear(fmt, data, w, h)))
    errors = []

    def work(k):
        fmt, w, h, data, (ok_o, want) = jobs[k]
        try:
            for rep in range(6):
                ok, out = hiplib.linear(fmt, data, w, h)
                assert ok == ok_o and np.array_equal(out, want), (fmt.name, rep)
                okb, outb = hiplib.block(fmt, data[:fmt.block_bytes])
                okc, wantb = oracle.blocks(fmt, data[:fmt.block_bytes])
                assert okb == bool(okc[0]) and (not okb or np.array_equal(outb, wantb[0])), (fmt.name, "leaf", rep)
                if not ok:      # this thread's own error text, not a neighbour's
                    assert hiplib.error() == "detexDecompressBlock: Decompress function for format 0x%08X returned error" % fmt.texture_format
        except Exception as e:  # noqa
            errors.append((names[k], repr(e)))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(len(jobs))]
    for t in threads: t.start()
    for t in threads: t.join()
    assert not errors, errors
