"""GPU tests of the host-pointer tier's band pipeline, the device-context rules and the multi-device C entry
(SURVEY.md 8b, 8e).  Everything goes through the C ABI; bit-exact against the oracle / the single-device call."""
import ctypes
import os

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


def _dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a).reshape(-1)).cuda()


@pytest.mark.parametrize("name,W,H", [("BC1", 8192, 8192), ("BPTC_FLOAT", 4096, 2048), ("RGTC1", 16384, 8192), ("BPTC", 4096, 4100), ("BPTC", 8192, 4100), ("ETC2", 2050, 9000)])
def test_host_tier_large_textures_match_device_tier(name, W, H, torch_cuda, hiplib):
    """large textures (64-256 MiB of pixels), incl. clipped geometry, on both sides of the duplex path's limit (32 MiB of blocks: BC1 8192^2, RGTC1
    16384 x 8192, BPTC 8192 x 4100 -- 1025 block rows in eight bands -- go up beside their download): the host-pointer drop-in entry == the device
    tier on the same stream, byte for byte, and the bool result agrees"""
    from detex_amd import binding
    torch = torch_cuda
    fmt = F.BY_NAME[name]
    wb, hb = (W + 3) // 4, (H + 3) // 4
    data = ol.stream_u(fmt, wb * hb, seed=0x9057 + fmt.index)
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    want = binding.decompress_linear_device(fmt, _dev(torch, data), W, H, status=status)
    torch.cuda.synchronize()
    ok, got = hiplib.linear(fmt, data, W, H)
    assert got.size == W * H * fmt.pixel_bytes
    assert np.array_equal(got, want.cpu().numpy())
    assert ok == bool(status.item() == 0)
    # block-major layout through the same pipeline
    if W % 4 == 0 and H % 4 == 0 and name in ("BC1", "BPTC_FLOAT"):
        want_t = binding.decompress_tiled_device(fmt, _dev(torch, data), wb, hb)
        torch.cuda.synchronize()
        ok_t, got_t = hiplib.tiled(fmt, data, wb, hb)
        assert np.array_equal(got_t, want_t.cpu().numpy()) and ok_t == ok


def test_host_tier_leaves_uncovered_pixels_untouched(torch_cuda, hiplib, oracle):
    """a block grid smaller than the image (width > 4*wb, height > 4*hb): the reference writes only covered pixels
    (texture.c:116-136); the caller's other bytes must survive"""
    fmt = F.BY_NAME["BC1"]
    W, H, wb, hb = 70, 45, 16, 10                  # grid covers 64 x 40
    data = ol.stream_u(fmt, wb * hb, seed=77)
    out = np.full(W * H * 4, 0xA5, np.uint8)
    ok, got = hiplib.linear(fmt, data, W, H, out=out, wb=wb, hb=hb)
    img = got.reshape(H, W, 4)
    _, want = oracle.linear(fmt, data, 64, 40)
    assert np.array_equal(img[:40, :64].reshape(-1), want)
    assert (img[40:] == 0xA5).all() and (img[:, 64:] == 0xA5).all()


def test_release_thread_resources_and_reuse(torch_cuda, hiplib, oracle):
    torch = torch_cuda
    fmt = F.BY_NAME["BC3"]
    W, H = 4096, 4096
    data = ol.stream_u(fmt, (W // 4) * (H // 4), seed=5)
    ok1, a = hiplib.linear(fmt, data, W, H)
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    hiplib.lib.detexhipReleaseThreadResources.restype = None
    hiplib.lib.detexhipReleaseThreadResources()
    free_after = torch.cuda.mem_get_info()[0]
    assert free_after - free_before >= W * H * 4            # the 64 MiB output staging buffer came back
    ok2, b = hiplib.linear(fmt, data, W, H)                  # the context is rebuilt on demand
    assert ok1 == ok2 and np.array_equal(a, b)
    _, want = oracle.linear(fmt, data[:64 * (W // 4) * 16], W, 256)
    assert np.array_equal(b[:want.size], want)


def test_device_tier_rejects_misaligned_pointers(torch_cuda):
    from detex_amd import binding
    torch = torch_cuda
    lib = binding.load()
    fmt = F.BY_NAME["BPTC"]
    buf = torch.zeros(64 * 16 + 64, dtype=torch.uint8, device="cuda")
    out = torch.zeros(64 * 64 + 64, dtype=torch.uint8, device="cuda")
    rc = lib.detexhipDecompressTextureLinearDevice(fmt.texture_format, buf.data_ptr() + 8, 32, 32, 8, 8, out.data_ptr(), 128, 0x334, None, None)
    assert rc != 0 and b"16-byte aligned" in lib.detexGetErrorMessage()
    rc = lib.detexhipDecompressTextureTiledDevice(fmt.texture_format, buf.data_ptr(), 8, 8, out.data_ptr() + 4, 0x334, None, None)
    assert rc != 0 and b"aligned" in lib.detexGetErrorMessage()
    rc = lib.detexhipDecompressTextureLinearDevice(fmt.texture_format, buf.data_ptr(), 32, 32, 8, 8, out.data_ptr(), 128, 0x334, None, None)
    assert rc == 0
    torch.cuda.synchronize()


def test_set_device_rules(torch_cuda):
    from detex_amd import binding
    lib = binding.load()
    n = lib.detexhipGetDeviceCount()
    assert n >= 1
    import torch
    before = torch.cuda.current_device()
    lib.detexhipReleaseThreadResources.restype = None
    lib.detexhipReleaseThreadResources()                     # this thread may have used device 0 in earlier tests
    assert lib.detexhipSetDevice(n + 3) != 0 and b"no such device" in lib.detexGetErrorMessage()
    assert lib.detexhipSetDevice(0) == 0
    out = np.zeros(16 * 4, np.uint8)
    blk = np.arange(8, dtype=np.uint8)
    lib.detexDecompressBlockBC1.restype = ctypes.c_bool
    assert lib.detexDecompressBlockBC1(blk.ctypes.data_as(ctypes.c_void_p), 0xFFFFFFFF, 0, out.ctypes.data_as(ctypes.c_void_p))
    # the context now lives on device 0: another device is refused BEFORE the thread's current device is touched
    assert lib.detexhipSetDevice(1) != 0 and (b"already used" in lib.detexGetErrorMessage() or b"no such device" in lib.detexGetErrorMessage())
    assert torch.cuda.current_device() == before             # selecting the host tier's device never moves the caller's


@pytest.mark.parametrize("name,W,H,shards", [("BC1", 4096, 4096, None), ("BC1", 2048, 1000, 3), ("BPTC_FLOAT", 1024, 2048, 5), ("BPTC", 1024, 1028, 2), ("RGTC1", 512, 36, 8)])
def test_multi_device_entry_matches_single_device(name, W, H, shards, torch_cuda, oracle):
    """detexhipDecompressTextureLinearMultiDevice with n = device count (1 on the test box) and with several shards
    placed on the same device: bands concatenated == the single-device call == the peer-gathered image; uploaded
    host blocks and device-resident blocks give the same result; status per shard"""
    from detex_amd import binding
    torch = torch_cuda
    fmt = F.BY_NAME[name]
    ndev = binding.load().detexhipGetDeviceCount()
    devices = list(range(ndev)) if shards is None else [g % ndev for g in range(shards)]
    wb, hb = (W + 3) // 4, (H + 3) // 4
    data = ol.stream_u(fmt, wb * hb, seed=0x3017 + fmt.index + len(devices))
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    want = binding.decompress_linear_device(fmt, _dev(torch, data), W, H, status=status).cpu().numpy()
    torch.cuda.synchronize()
    r = binding.decompress_linear_multi_device(fmt, W, H, devices, host_blocks=data, gather_device=0)
    got = np.concatenate([b.cpu().numpy()[:max(0, min(s[1] * 4, H) - s[0] * 4) * W * fmt.pixel_bytes] for b, s in zip(r["bands"], r["shards"])])
    assert np.array_equal(got, want)
    assert np.array_equal(r["gathered"].cpu().numpy(), want)
    assert r["ok"] == bool(status.item() == 0)
    assert [s[0] for s in r["shards"]][0] == 0 and r["shards"][-1][1] == hb
    assert all(a[1] == b[0] for a, b in zip(r["shards"], r["shards"][1:]))
    assert r["decode_wall_ms"] > 0 and all(s[2] >= 0 for s in r["shards"])
    # peer mapping towards the gather device, per shard: same device -> 1; another device -> whatever the topology gives, reported not hidden
    assert all(p == 1 for p, d in zip(r["peer_access"], devices) if d == 0) and all(p in (0, 1) for p in r["peer_access"])
    if ndev > 1:        # gathered on the LAST device instead: every other shard crosses a link
        r3 = binding.decompress_linear_multi_device(fmt, W, H, devices, host_blocks=data, gather_device=ndev - 1)
        assert np.array_equal(r3["gathered"].cpu().numpy(), want) and all(p in (0, 1) for p in r3["peer_access"])
    # blocks already resident on the devices
    dblocks = []
    for g, dev in enumerate(devices):
        r0, r1 = r["shards"][g][0], r["shards"][g][1]
        chunk = data[r0 * wb * fmt.block_bytes:r1 * wb * fmt.block_bytes]
        dblocks.append(torch.from_numpy(np.ascontiguousarray(chunk) if chunk.size else np.zeros(16, np.uint8)).to("cuda:%d" % dev))
    r2 = binding.decompress_linear_multi_device(fmt, W, H, devices, device_blocks=dblocks)
    got2 = np.concatenate([b.cpu().numpy()[:max(0, min(s[1] * 4, H) - s[0] * 4) * W * fmt.pixel_bytes] for b, s in zip(r2["bands"], r2["shards"])])
    assert np.array_equal(got2, want) and r2["gathered"] is None and all(p == -1 for p in r2["peer_access"])
    # a sample of the result against the CPU oracle, so the comparison above is not GPU against GPU only
    rows = min(hb, 8)
    _, ref_rows = oracle.linear(fmt, data[:rows * wb * fmt.block_bytes], W, min(rows * 4, H))
    assert np.array_equal(want[:ref_rows.size], ref_rows)


@pytest.mark.parametrize("name,W,H,shards,pad", [("BC1", 1024, 520, 3, 48), ("BPTC_FLOAT", 512, 256, 4, 64), ("RGTC1", 512, 36, 2, 7)])
def test_multi_device_gather_with_padded_pitch(name, W, H, shards, pad, torch_cuda, oracle):
    """pitch_bytes > width * pixel_size: a band is (rows - 1) * pitch + width * px bytes, so the gather must go row by row
    and leave the bytes between the rows alone in the bands and in the gathered image alike (canaries)"""
    from detex_amd import binding
    fmt = F.BY_NAME[name]
    px = fmt.pixel_bytes
    unit = px if px < 4 else 4
    pitch = W * px + (pad + unit - 1) // unit * unit
    ndev = binding.load().detexhipGetDeviceCount()
    devices = [g % ndev for g in range(shards)]
    wb, hb = (W + 3) // 4, (H + 3) // 4
    data = ol.stream_u(fmt, wb * hb, seed=0x91C4 + fmt.index)
    _, want = oracle.linear(fmt, data, W, H)
    r = binding.decompress_linear_multi_device(fmt, W, H, devices, host_blocks=data, gather_device=0, pitch=pitch)
    img = r["gathered"].cpu().numpy().reshape(H, pitch)
    assert np.array_equal(img[:, :W * px].reshape(-1), want.reshape(-1))
    assert (img[:, W * px:] == 0xA5).all(), "the gather wrote between the rows of the gathered image"
    for band, s in zip(r["bands"], r["shards"]):
        rows = max(0, min(s[1] * 4, H) - s[0] * 4)
        b = band.cpu().numpy()[:rows * pitch].reshape(rows, pitch)
        assert np.array_equal(b[:, :W * px].reshape(-1), want.reshape(H, W * px)[s[0] * 4:s[0] * 4 + rows].reshape(-1))
        assert (b[:, W * px:] == 0xA5).all()


@pytest.mark.parametrize("name,W,H,shards", [("BC1", 2048, 2048, 4), ("BPTC", 1022, 515, 3), ("BPTC_FLOAT", 1024, 512, 2), ("RGTC1", 516, 40, 8), ("BC1", 64, 64, 1)])
def test_multi_device_host_entry(name, W, H, shards, torch_cuda, oracle, hiplib):
    """detexhipDecompressTextureLinearMultiDeviceHost: host blocks in, host pixels out, one worker per shard (all on the
    devices present: one on the test box) == the oracle == the single-device host tier, incl. clipped sizes, a padded pitch
    with canaries, and the reference's bool result"""
    from detex_amd import binding
    fmt = F.BY_NAME[name]
    px = fmt.pixel_bytes
    ndev = binding.load().detexhipGetDeviceCount()
    devices = [g % ndev for g in range(shards)]
    wb, hb = (W + 3) // 4, (H + 3) // 4
    data = ol.stream_u(fmt, wb * hb, seed=0x405B + fmt.index)
    ok_o, want = oracle.linear(fmt, data, W, H)
    ok, got, wall = binding.decompress_linear_multi_device_host(fmt, data, W, H, devices)
    assert ok == ok_o and np.array_equal(got, want.reshape(-1)) and wall > 0
    ok_h, got_h = hiplib.linear(fmt, data, W, H)
    assert ok_h == ok and np.array_equal(got_h.reshape(-1), got)
    pitch = W * px + 4 * (px if px < 4 else 4)
    canvas = np.full(H * pitch, 0xA5, np.uint8)
    ok2, _, _ = binding.decompress_linear_multi_device_host(fmt, data, W, H, devices, out=canvas, pitch=pitch)
    img = canvas.reshape(H, pitch)
    assert ok2 == ok_o and np.array_equal(img[:, :W * px].reshape(-1), want.reshape(-1)) and (img[:, W * px:] == 0xA5).all()
    if not ok_o:
        assert b"returned error" in binding.load().detexGetErrorMessage()


def test_multi_device_entries_from_concurrent_threads(torch_cuda, oracle):
    """the multi-device entries keep their streams / buffers per calling thread and take no lock: several threads calling
    at once get their own results"""
    import threading
    from detex_amd import binding
    binding.load()
    fmt = F.BY_NAME["BC3"]
    errors = []

    def worker(t):
        try:
            W, H = 512 + 64 * t, 256
            data = ol.stream_u(fmt, (W // 4) * (H // 4), seed=0x7700 + t)
            _, want = oracle.linear(fmt, data, W, H)
            for _ in range(4):
                r = binding.decompress_linear_multi_device(fmt, W, H, [0, 0, 0], host_blocks=data, gather_device=0)
                if not np.array_equal(r["gathered"].cpu().numpy(), want.reshape(-1)):
                    errors.append(("device entry", t))
                ok, got, _ = binding.decompress_linear_multi_device_host(fmt, data, W, H, [0, 0])
                if not np.array_equal(got, want.reshape(-1)):
                    errors.append(("host entry", t))
            lib = binding.load()
            lib.detexhipReleaseThreadResources.restype = None
            lib.detexhipReleaseThreadResources()
        except Exception as e:  # noqa
            errors.append((t, repr(e)))
    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


def test_duplex_staged_path_from_concurrent_threads(torch_cuda, hiplib):
    """three threads, each decoding its own large texture (RGTC2 8192 x 4100 / 8192 x 4104 / 8192 x 4108: 32+ MiB of blocks, so the upload runs beside
    the download on a helper thread and a second stream per calling thread) twice at the same time: every thread gets its own pixels, equal
    to the device tier's on the same blocks"""
    import threading
    from detex_amd import binding
    torch = torch_cuda
    fmt = F.BY_NAME["RGTC2"]
    cases = []
    for t in range(3):
        W, H = 8192, 4100 + 4 * t
        data = ol.stream_u(fmt, (W // 4) * (H // 4), seed=0xD0B1 + t)
        want = binding.decompress_linear_device(fmt, _dev(torch, data), W, H).cpu().numpy()
        cases.append((W, H, data, want))
    torch.cuda.synchronize()
    errors = []

    def worker(t):
        try:
            W, H, data, want = cases[t]
            for _ in range(2):
                ok, got = hiplib.linear(fmt, data, W, H)
                if not ok or not np.array_equal(got, want):
                    errors.append(("pixels differ", t))
            hiplib.lib.detexhipReleaseThreadResources.restype = None
            hiplib.lib.detexhipReleaseThreadResources()
        except Exception as e:  # noqa
            errors.append((t, repr(e)))
    threads = [threading.Thread(target=worker, args=(t,)) for t in range(3)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


# ---- the gather's three copy branches (multi_device.cpp: flat peer copy, 2-D copy over a peer mapping, row-wise copies without one) ----------
_GATHER_WORKER = r'''
import json, sys
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import oracle_lib as ol
from detex_amd import binding, formats as F
binding.load()
orc = ol.Oracle()
ndev = binding.load().detexhipGetDeviceCount()
out = {"ndev": ndev, "cases": []}
for name, W, H, shards, pad in (("BC1", 1024, 520, 3, 0), ("BC1", 1024, 520, 3, 48), ("BPTC_FLOAT", 512, 256, 4, 64), ("RGTC1", 512, 36, 2, 7)):
    fmt = F.BY_NAME[name]
    px = fmt.pixel_bytes
    unit = px if px < 4 else 4
    pitch = W * px + (pad + unit - 1) // unit * unit
    data = ol.stream_u(fmt, ((W + 3) // 4) * ((H + 3) // 4), seed=0x6A7 + fmt.index + pad)
    _, want = orc.linear(fmt, data, W, H)
    for gather in sorted({0, ndev - 1}):
        devices = [(gather + 1 + g) %% ndev for g in range(shards)]       # with >= 2 devices the first shard is NOT on the gather device
        r = binding.decompress_linear_multi_device(fmt, W, H, devices, host_blocks=data, gather_device=gather, pitch=pitch if pad else None)
        img = r["gathered"].cpu().numpy().reshape(H, pitch)
        out["cases"].append({"name": name, "pad": pad, "gather": gather, "devices": devices, "peer_access": r["peer_access"],
                             "pixels": bool(np.array_equal(img[:, :W * px].reshape(-1), want.reshape(-1))),
                             "canaries": bool((img[:, W * px:] == 0xA5).all())})
print(json.dumps(out))
'''


def _run_gather_worker(env_extra):
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _GATHER_WORKER % {"root": root}], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env_extra))
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_gather_branch_without_peer_mapping_runs_on_any_box():
    """DETEXHIP_PEER_ACCESS=0 (multi_device.cpp: peer_mapping_allowed): no pair is treated as mapped, so every gather copy takes the branch
    of an unmapped pair -- flat hipMemcpyPeerAsync for dense rows, ONE hipMemcpyPeerAsync PER ROW for padded pitches -- which a one-GPU box
    otherwise never executes (a device always 'maps' itself).  Pixels == the oracle, canaries between the rows intact, peer_access == 0"""
    res = _run_gather_worker({"DETEXHIP_PEER_ACCESS": "0"})
    assert len(res["cases"]) >= 4
    for c in res["cases"]:
        assert c["pixels"] and c["canaries"], c
        assert all(p == 0 for p in c["peer_access"]), c


def test_gather_across_devices_all_three_branches():
    """gather_device != the shard's device (needs >= 2 visible GPUs): dense rows (flat peer copy), padded rows over a peer mapping (2-D copy)
    and -- in the test above -- padded rows without one.  On a one-GPU box this is SKIPPED, and says so: the cross-device copies of
    multi_device.cpp have then only run with source and destination on the same device"""
    import torch
    if torch.cuda.device_count() < 2:
        reason = ("only %d GPU visible: the cross-device gather copies (hipMemcpyPeerAsync / hipMemcpy2DAsync between two devices) cannot execute here; "
                  "same-device and no-mapping branches are covered by the tests above" % torch.cuda.device_count())
        print("SKIPPED:", reason)
        pytest.skip(reason)
    res = _run_gather_worker({})
    crossed = 0
    for c in res["cases"]:
        assert c["pixels"] and c["canaries"], c
        crossed += sum(1 for d in c["devices"] if d != c["gather"])
        assert all(p in (0, 1) for p in c["peer_access"]), c
    assert crossed > 0
    print("peer_access per case:", [(c["name"], c["pad"], c["gather"], c["peer_access"]) for c in res["cases"]])


@pytest.mark.parametrize("W,H", [(1024, 1024), (4100, 4096), (512, 512), (8192, 4100), (2048, 2048), (2048, 2052)])
def test_host_tier_status_does_not_leak_between_calls(W, H, hiplib, oracle):
    """a texture with invalid blocks (result false) followed by a valid one of the same size (result true) and again an invalid one, on
    the pinned-exchange path (512^2), the staged path with its status word in pinned memory (1024^2) and with the device word that is
    kept zero between calls (> 2^18 blocks: both sides of that limit, 2048^2 and 2048 x 2052), and the duplex staged path (8192 x 4100: 32 MiB of blocks in
    whole blocks: uploads beside downloads, eight bands sharing the device word): the status of one call never shows in the next"""
    fmt = F.BY_NAME["BPTC"]
    wb, hb = (W + 3) // 4, (H + 3) // 4
    bad = ol.stream_u(fmt, wb * hb, seed=0x57A7)                 # random BC7: 0.4 % reserved blocks
    good = bad.copy().reshape(-1, 16)
    good[:, 0] |= 1                                             # every block mode 0: valid
    good = good.reshape(-1)
    rows = min(hb, 16)
    for k, (data, want_ok) in enumerate([(bad, False), (good, True), (bad, False), (good, True), (good, True)]):
        ok, got = hiplib.linear(fmt, data, W, H)
        assert ok == want_ok, (W, H, k)
        _, ref_rows = oracle.linear(fmt, data[:rows * wb * 16], W, min(rows * 4, H))
        assert np.array_equal(got[:ref_rows.size], ref_rows), (W, H, k)
        if not ok:
            assert hiplib.error() == "detexDecompressBlock: Decompress function for format 0x%08X returned error" % fmt.texture_format


# ---- mid-size textures (0.25 - 2 MiB of pixels): the banded pinned exchange and the lower end of the staged path (host_tier.cpp) -------------
@pytest.mark.parametrize("name,W,H,wb,hb", [("BC1", 512, 512, None, None), ("BPTC", 724, 724, None, None), ("BPTC", 701, 333, None, None), ("BC3", 1001, 513, None, None),
                                            ("RGTC1", 2048, 1024, None, None), ("BPTC_FLOAT", 512, 384, None, None), ("ETC2", 900, 400, 200, 90), ("EAC_RG11", 1022, 258, None, None)])
def test_host_tier_mid_size_window(name, W, H, wb, hb, hiplib, oracle):
    """0.25-2 MiB of pixels, both sides of the pinned exchange's 1.25 MiB limit: whole and clipped sizes, a block grid smaller than the image
    (the rest of the caller's buffer must survive), invalid blocks (zero-filled, false + the reference's text), 64-bit pixels, narrow
    pixels -- twice each"""
    fmt = F.BY_NAME[name]
    gwb, ghb = (W + 3) // 4 if wb is None else wb, (H + 3) // 4 if hb is None else hb
    data = ol.stream_u(fmt, gwb * ghb, seed=0x4E6 + fmt.index + W)
    px = fmt.pixel_bytes
    assert 256 << 10 < W * H * px <= 2100 << 10
    for rep in range(2):
        out = np.full(W * H * px + 64, 0xA5, np.uint8)
        ok, got = hiplib.linear(fmt, data, W, H, out=out[:W * H * px], wb=gwb, hb=ghb)
        cw, ch = min(W, 4 * gwb), min(H, 4 * ghb)
        want_ok, want = oracle.linear(fmt, data, cw, ch)                 # (cw, ch: the part of the image the block grid covers)
        img = got.reshape(H, W * px)
        assert np.array_equal(img[:ch, :cw * px].reshape(-1), want.reshape(-1)), (name, W, H, rep)
        assert (img[:ch, cw * px:] == 0xA5).all() and (img[ch:] == 0xA5).all(), "pixels outside the block grid were written"
        assert (out[W * H * px:] == 0xA5).all(), "wrote past the image"
        assert ok == want_ok
        if not ok:
            assert hiplib.error() == "detexDecompressBlock: Decompress function for format 0x%08X returned error" % fmt.texture_format


def test_host_tier_unusual_caller_buffers(hiplib, oracle, torch_cuda):
    """a pixel buffer that is pinned by the caller (a torch pinned tensor), one that is not 16-byte aligned, and four threads decoding into
    adjacent images inside one allocation (neighbours share a page at every boundary): the result is the same"""
    import threading
    torch = torch_cuda
    fmt = F.BY_NAME["BC1"]
    W = H = 512
    data = ol.stream_u(fmt, (W // 4) * (H // 4), seed=0xFA11)
    _, want = oracle.linear(fmt, data, W, H)
    pinned = torch.empty(W * H * 4, dtype=torch.uint8).pin_memory()
    ok, got = hiplib.linear(fmt, data, W, H, out=pinned.numpy())
    assert ok and np.array_equal(got, want)
    raw = np.full(W * H * 4 + 16, 0xA5, np.uint8)
    ok, got = hiplib.linear(fmt, data, W, H, out=raw[4:4 + W * H * 4])             # misaligned by 4: the staged-rows kernel
    assert ok and np.array_equal(got, want) and (raw[:4] == 0xA5).all() and (raw[4 + W * H * 4:] == 0xA5).all()
    # four threads, adjacent images inside ONE allocation (neighbours share a page at every boundary), 40 calls each
    n = 4
    sz = 500 * 300 * 4                                                               # not a multiple of the page size
    big = np.zeros(n * sz, np.uint8)
    d2 = ol.stream_u(fmt, 125 * 75, seed=0xFA12)
    _, want2 = oracle.linear(fmt, d2, 500, 300)
    errors = []

    def worker(k):
        try:
            api = ol.DetexAPI(hiplib.path)
            for _ in range(40):
                view = big[k * sz:(k + 1) * sz]
                view[:] = 0
                ok, got = api.linear(fmt, d2, 500, 300, out=view)
                if not ok or not np.array_equal(got, want2):
                    errors.append((k, "mismatch"))
                    return
        except Exception as e:   # noqa
            errors.append((k, repr(e)))
    threads = [threading.Thread(target=worker, args=(k,)) for k in range(n)]
    for t in threads: t.start()
    for t in threads: t.join()
    assert not errors, errors


@pytest.mark.parametrize("W,H,wb,hb", [(512, 300, 128, 128), (304, 64, 76, 40), (1024, 1024, 256, 300), (64, 20, 16, 9)])
def test_block_grid_taller_than_the_image(W, H, wb, hb, hiplib, oracle, ref):
    """a block grid that reaches below the image (height < 4 * height_in_blocks by more than a block row): the reference decodes every block of
    the grid, stores only what lies inside the image, and returns false if ANY block of the grid is invalid (texture.c:112-144) -- through
    the resident service, the banded pinned exchange and the staged path alike.  (A grid WIDER than the image by more than a block makes the
    reference itself copy a negative number of bytes: outside its contract, not tested.)"""
    fmt = F.BY_NAME["BPTC"]
    data = ol.stream_u(fmt, wb * hb, seed=0x7A11 + W)
    blk = data.reshape(-1, 16)
    blk[:, 0] |= 1                                   # every block valid ...
    ok_all, want = oracle.linear(fmt, data, W, H, wb=wb, hb=hb)
    assert ok_all
    if ref is not None:
        ok_r, want_r = ref.linear(fmt, data, W, H, wb=wb, hb=hb)
        assert ok_r and np.array_equal(want_r, want)
    for rep in range(2):
        ok, got = hiplib.linear(fmt, data, W, H, wb=wb, hb=hb)
        assert ok and np.array_equal(got, want), (W, H, rep)
    blk[(hb - 1) * wb + 3, :] = 0                     # ... except one in the LAST block row, which lies wholly below the image
    ok_bad, want_bad = oracle.linear(fmt, data, W, H, wb=wb, hb=hb)
    assert not ok_bad and np.array_equal(want_bad, want)
    for rep in range(2):
        ok, got = hiplib.linear(fmt, data, W, H, wb=wb, hb=hb)
        assert not ok and np.array_equal(got, want), (W, H, rep)


# ---- pixel buffers handed out by the library (detexhipAllocPixelBuffer): the kernel writes straight into them --------------------------------
def _owned(lib, nbytes):
    lib.detexhipAllocPixelBuffer.restype = ctypes.c_void_p
    lib.detexhipAllocPixelBuffer.argtypes = [ctypes.c_size_t]
    lib.detexhipFreePixelBuffer.argtypes = [ctypes.c_void_p]
    lib.detexhipFreePixelBuffer.restype = None
    p = lib.detexhipAllocPixelBuffer(nbytes)
    assert p, lib.detexGetErrorMessage()
    return p, np.ctypeslib.as_array((ctypes.c_uint8 * nbytes).from_address(p))


@pytest.mark.parametrize("name,W,H,wb,hb", [("BC1", 512, 512, None, None), ("BPTC", 1024, 1024, None, None), ("BPTC", 701, 333, None, None), ("BC3", 1401, 1399, None, None),
                                            ("RGTC1", 2048, 1024, None, None), ("BPTC_FLOAT", 512, 384, None, None), ("ETC2", 900, 400, 200, 90), ("BC1", 2048, 2048, None, None),
                                            ("BC1", 64, 64, None, None), ("ETC2_EAC", 300, 200, None, None)])
def test_library_owned_pixel_buffers(name, W, H, wb, hb, hiplib, oracle):
    """detexDecompressTextureLinear into a sub-range of a buffer from detexhipAllocPixelBuffer: the direct-write path (up to 8 MiB of pixels:
    whole and clipped sizes, a grid smaller than the image, invalid blocks, 64-bit and narrow pixels), the resident service (64^2), the staged
    path downloading into pinned memory (2048^2) -- pixels == the oracle, bytes around the image untouched, twice each"""
    fmt = F.BY_NAME[name]
    gwb, ghb = (W + 3) // 4 if wb is None else wb, (H + 3) // 4 if hb is None else hb
    data = ol.stream_u(fmt, gwb * ghb, seed=0x0B0F + fmt.index + W)
    px = fmt.pixel_bytes
    n = W * H * px
    ptr, whole = _owned(hiplib.lib, n + 4096 + 160)
    try:
        for rep, off in enumerate((64, 4096 + 16)):                              # the image anywhere inside the allocation
            whole[:] = 0xA5
            ok, got = hiplib.linear(fmt, data, W, H, out=whole[off:off + n], wb=gwb, hb=ghb)
            cw, ch = min(W, 4 * gwb), min(H, 4 * ghb)
            want_ok, want = oracle.linear(fmt, data, cw, ch)
            img = got.reshape(H, W * px)
            assert np.array_equal(img[:ch, :cw * px].reshape(-1), want.reshape(-1)), (name, W, H, rep)
            assert (img[:ch, cw * px:] == 0xA5).all() and (img[ch:] == 0xA5).all(), "pixels outside the block grid were written"
            assert (whole[:off] == 0xA5).all() and (whole[off + n:] == 0xA5).all(), "wrote outside the image"
            assert ok == want_ok
            if not ok:
                assert hiplib.error() == "detexDecompressBlock: Decompress function for format 0x%08X returned error" % fmt.texture_format
        # the BLOCKS in owned memory as well (16-byte aligned inside it): read by the kernel where they are
        pb, blocks_view = _owned(hiplib.lib, data.size + 64)
        try:
            blocks_view[16:16 + data.size] = data
            whole[:] = 0xA5
            ok, got = hiplib.linear(fmt, blocks_view[16:16 + data.size], W, H, out=whole[64:64 + n], wb=gwb, hb=ghb)
            assert ok == want_ok and np.array_equal(got.reshape(H, W * px)[:ch, :cw * px].reshape(-1), want.reshape(-1))
            blocks_view[3:3 + data.size] = data                                    # ... and misaligned inside it: copied like any other host pointer
            ok, got = hiplib.linear(fmt, blocks_view[3:3 + data.size], W, H, out=whole[64:64 + n], wb=gwb, hb=ghb)
            assert ok == want_ok and np.array_equal(got.reshape(H, W * px)[:ch, :cw * px].reshape(-1), want.reshape(-1))
        finally:
            hiplib.lib.detexhipFreePixelBuffer(pb)
    finally:
        hiplib.lib.detexhipFreePixelBuffer(ptr)


def test_library_owned_pixel_buffers_lifecycle(hiplib, oracle):
    """freeing twice, freeing a foreign pointer (refused with a message, nothing freed), NULL (ignored); a buffer that has been freed is an
    ordinary pointer again; two threads decoding into their own owned buffers"""
    import threading
    lib = hiplib.lib
    fmt = F.BY_NAME["BC1"]
    p, view = _owned(lib, 1 << 20)
    lib.detexhipFreePixelBuffer(p)
    lib.detexSetErrorMessage(b"(none)")
    lib.detexhipFreePixelBuffer(p)                                                 # second free: refused
    assert "was not returned by detexhipAllocPixelBuffer" in hiplib.error()
    lib.detexhipFreePixelBuffer(None)
    data = ol.stream_u(fmt, 128 * 128, seed=3)
    _, want = oracle.linear(fmt, data, 512, 512)
    ok, got = hiplib.linear(fmt, data, 512, 512)                                    # an ordinary numpy buffer afterwards: the other paths
    assert ok and np.array_equal(got, want)
    errors = []

    def worker(k):
        try:
            api = ol.DetexAPI(hiplib.path)
            ptr, buf = _owned(api.lib, 512 * 512 * 4)
            for _ in range(50):
                buf[:] = 0
                ok, got = api.linear(fmt, data, 512, 512, out=buf)
                if not ok or not np.array_equal(got, want):
                    errors.append((k, "mismatch")); break
            api.lib.detexhipFreePixelBuffer(ptr)
        except Exception as e:   # noqa
            errors.append((k, repr(e)))
    th = [threading.Thread(target=worker, args=(k,)) for k in range(3)]
    for t in th: t.start()
    for t in th: t.join()
    assert not errors, errors


# ---- round 6: lifetime and failure hazards of the host tier ------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,W,H", [("BC1", 512, 512), ("BPTC_FLOAT", 256, 256), ("RGTC2", 300, 200)])
def test_owned_pixel_buffer_at_odd_offsets(name, W, H, hiplib, oracle):
    """a pixel buffer that lies inside library-owned memory but is NOT aligned to the target pixel (owned + 1, + 2, + 6): the kernel's row
    stores cannot write there, so the call takes the copying paths like any other pointer -- same pixels, nothing outside the image touched"""
    fmt = F.BY_NAME[name]
    data = ol.stream_u(fmt, ((W + 3) // 4) * ((H + 3) // 4), seed=0x0DD + W)
    n = W * H * fmt.pixel_bytes
    want_ok, want = oracle.linear(fmt, data, W, H)
    ptr, whole = _owned(hiplib.lib, n + 256)
    try:
        for off in (1, 2, 6, 64 + 3):
            whole[:] = 0x5A
            ok, got = hiplib.linear(fmt, data, W, H, out=whole[off:off + n])
            assert ok == want_ok and np.array_equal(got, want), (name, off)
            assert (whole[:off] == 0x5A).all() and (whole[off + n:] == 0x5A).all(), (name, off)
    finally:
        hiplib.lib.detexhipFreePixelBuffer(ptr)


def test_free_of_an_owned_buffer_waits_for_the_decode_in_flight(hiplib, oracle):
    """detexhipFreePixelBuffer from one thread while another thread's kernel is writing into that buffer: the free waits for the decode (the
    decode holds the buffer from the lookup to its completion), the decode returns true with nothing lost, and the buffer is freed exactly
    once -- 25 rounds, the free issued at varying moments after the decode call has started"""
    import threading, time
    lib = hiplib.lib
    fmt = F.BY_NAME["BPTC"]
    W = H = 1400                                                     # 7.5 MiB of pixels: the direct-write path, ~0.4 ms per call
    data = ol.stream_u(fmt, 350 * 350, seed=0xF3EE).reshape(-1, 16)
    data[:, 0] |= 1
    data = data.reshape(-1)
    _, want = oracle.linear(fmt, data, W, H)
    want_sum = int(want.astype(np.uint64).sum())
    results = []
    for rnd in range(25):
        ptr, view = _owned(lib, W * H * 4)
        started = threading.Event()
        seen = {}

        def decoder():
            api = ol.DetexAPI(hiplib.path)
            api.linear(fmt, data, W, H, out=view)                     # (this thread's context exists before the timed call)
            view[:] = 0
            started.set()
            ok, got = api.linear(fmt, data, W, H, out=view)
            seen["returned"] = time.perf_counter()
            seen["ok"] = ok
        t = threading.Thread(target=decoder)
        t.start()
        started.wait()
        time.sleep(0.00002 * (rnd % 8))                               # 0 .. 140 us into the call
        lib.detexhipFreePixelBuffer(ptr)
        freed = time.perf_counter()
        t.join()
        assert seen["ok"], hiplib.error()
        results.append(freed - seen["returned"])
        lib.detexSetErrorMessage(b"(none)")
        lib.detexhipFreePixelBuffer(ptr)                              # freed exactly once
        assert "was not returned by detexhipAllocPixelBuffer" in hiplib.error()
    # and the memory is an ordinary pointer afterwards: a fresh owned buffer decodes correctly
    ptr, view = _owned(lib, W * H * 4)
    try:
        ok, got = hiplib.linear(fmt, data, W, H, out=view)
        assert ok and int(got.astype(np.uint64).sum()) == want_sum and np.array_equal(got, want)
    finally:
        lib.detexhipFreePixelBuffer(ptr)
    print("free returned %.0f .. %.0f us relative to the decode's return" % (min(results) * 1e6, max(results) * 1e6))


@pytest.mark.parametrize("path,name,W,H", [("leaf", "BPTC", 4, 4), ("pinned", "BPTC", 256, 256), ("banded", "BPTC", 512, 512), ("staged_pinned_status", "BPTC", 1024, 1024),
                                           ("staged_device_status", "BPTC", 4100, 4096), ("staged_duplex", "BPTC", 8192, 4100), ("owned", "BPTC", 1024, 1024), ("tiled_pinned", "BPTC", 256, 256),
                                           ("blocks_direct", "BPTC", 0, 0), ("blocks_staged", "BPTC", 0, 0), ("histogram", "BPTC", 0, 0)])
def test_call_after_a_failure_behind_the_launch_starts_clean(path, name, W, H, hiplib, oracle):
    """detexhipTestFailAfterLaunch(1): the next call returns false right behind its kernel launch -- with invalid blocks in its input, so the
    kernels it leaves running RAISE the status word, count on the completion counters and write into the pinned exchange buffer.  The
    call after it (all blocks valid) must return true with the oracle's pixels on every host path: the thread's stream is drained and its
    device words are zeroed before anything new is launched (host_tier.cpp: heal_if_dirty)."""
    lib = hiplib.lib
    lib.detexhipTestFailAfterLaunch.argtypes = [ctypes.c_int]
    lib.detexhipTestFailAfterLaunch.restype = None
    fmt = F.BY_NAME[name]
    if path in ("blocks_direct", "blocks_staged", "histogram"):
        n = 3000 if path == "blocks_direct" else (200000 if path == "blocks_staged" else 50000)
        bad = ol.stream_u(fmt, n, seed=0xBADB10C)
        good = bad.copy().reshape(-1, 16); good[:, 0] |= 1; good = good.reshape(-1)
        if path == "histogram":
            f = lib.detexhipModeHistogram
            f.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
            f.restype = ctypes.c_bool
            hist = np.zeros(16, np.uint32)
            assert f(fmt.texture_format, good.ctypes.data, n, hist.ctypes.data) and int(hist.sum()) == n
            first = hist.copy()
            lib.detexhipTestFailAfterLaunch(1)
            ok, _, _ = hiplib.blocks(fmt, bad)                          # fails behind its launch, status word raised by the kernel
            assert not ok and "injected failure" in hiplib.error()
            assert f(fmt.texture_format, good.ctypes.data, n, hist.ctypes.data) and np.array_equal(hist, first)
            return
        lib.detexhipTestFailAfterLaunch(1)
        ok, _, _ = hiplib.blocks(fmt, bad)
        assert not ok and "injected failure" in hiplib.error()
        ok, okb, px = hiplib.blocks(fmt, good)
        assert ok and (okb == 1).all()
        for i in (0, n // 2, n - 1):
            _, want = oracle.linear(fmt, good[16 * i:16 * i + 16], 4, 4)
            assert np.array_equal(px[i], want), (path, i)
        return
    wb, hb = (W + 3) // 4, (H + 3) // 4
    bad = ol.stream_u(fmt, wb * hb, seed=0xFA11 + W)
    bad[:16] = 0                                                      # the first block is reserved mode 8: invalid for sure
    good = bad.copy().reshape(-1, 16); good[:, 0] |= 1; good = good.reshape(-1)
    rows = min(hb, 8)
    _, want_rows = oracle.linear(fmt, good[:rows * wb * 16], W, min(rows * 4, H))
    owned = None
    if path == "owned":
        owned = _owned(lib, W * H * 4)

    def call(data):
        if path == "leaf":
            return hiplib.block(fmt, data[:16])
        if path == "tiled_pinned":
            return hiplib.tiled(fmt, data, wb, hb)
        return hiplib.linear(fmt, data, W, H, out=owned[1] if owned else None)
    try:
        for rnd in range(2):
            lib.detexhipTestFailAfterLaunch(1)
            ok, _ = call(bad)
            assert not ok and "injected failure" in hiplib.error(), (path, rnd, hiplib.error())
            ok, got = call(good)
            assert ok, (path, rnd, hiplib.error())
            if path == "leaf":
                _, want = oracle.linear(fmt, good[:16], 4, 4)
                assert np.array_equal(got, want)
            elif path == "tiled_pinned":
                _, want = oracle.linear(fmt, good[:16], 4, 4)
                assert np.array_equal(got[:64], want)
            else:
                assert np.array_equal(got[:want_rows.size], want_rows), (path, rnd)
            ok, _ = call(bad)                                         # and the status still works afterwards
            assert not ok and (path == "leaf" or "returned error" in hiplib.error())   # (the leaf functions set no message: decompress-*.c)
            ok, _ = call(good)
            assert ok
    finally:
        if owned:
            lib.detexhipFreePixelBuffer(owned[0])
