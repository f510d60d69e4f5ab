#!/usr/bin/env python3
"""Generate tests/golden/* from the COMPILED REFERENCE (oracle/_ref/libdetex_ref.so).

Runs only in the build container (needs /root/reference for the fixtures and oracle/_ref).
What it writes is data only -- inputs and the reference's outputs:

  tests/golden/test-texture-*.ktx   the 17 compressed 64x64 fixtures the reference bundles
                                    (copied verbatim; /root/reference/LICENSE permits)
  tests/golden/fixtures.json        sha256 of detexDecompressTextureLinear(fixture) per target format
  tests/golden/forced_vectors.npz   mode-forced blocks (tests/streams.py) + reference output + ok flag
  tests/golden/maskflags.json       per (format, mode_mask, flags): ok bitmap + sha256 of outputs
  tests/golden/clip.npz             clipped (width/height not multiples of 4) linear decodes
  tests/golden/digests_8192.json    sha256 (+ FNV-1a-64) of the reference output on the 8192x8192 streams U (all formats),
                                    M (BPTC, BPTC_FLOAT) and C (every format with a fixture), on converted targets, and on
                                    whole 32768-wide bands of the sharded configs ("bands": the first band of each; "bands_all": every band
                                    of BC1 32768^2 over 4 ranks and BPTC_FLOAT 32768^2 over 8, from the global stream offsets)
                                    ("clipped": large textures whose width / height are not multiples of four)
usage: python tools/make_goldens.py [fixtures] [vectors] [digests] [digests_c] [digests_pf] [bands] [bands_all] [clipped]   (default: all)
"""
import ctypes, hashlib, json, os, shutil, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from detex_amd import formats as F
import oracle_lib as ol, streams
from detex_amd.ktx import read_ktx

REF_DIR = "/root/reference"
G = os.path.join(ROOT, "tests", "golden")
sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()

def ref_linear_mt(ref, f, data, W, H, pf=None, threads=None):
    """detexDecompressTextureLinear of the compiled reference over row bands on a thread pool (all reference state is
    __thread; ctypes releases the GIL).  Returns (ok, pixels)."""
    from concurrent.futures import ThreadPoolExecutor
    threads = threads or min(16, os.cpu_count() or 1)
    pf = (f.texture_format & 0xFFFF) if pf is None else pf
    px = 1 + ((pf & 0xF00) >> 8)
    wb, hb = W // 4, H // 4
    out = np.empty(W * H * px, np.uint8)
    data = np.ascontiguousarray(data)

    def band(g):
        r0, r1 = g * hb // threads, (g + 1) * hb // threads
        if r1 <= r0:
            return True
        tex = ol.DetexTexture(f.texture_format, ol._ptr(data[r0 * wb * f.block_bytes:]), W, (r1 - r0) * 4, wb, r1 - r0)
        return bool(ref.lib.detexDecompressTextureLinear(ctypes.byref(tex), ol._ptr(out[r0 * 4 * W * px:]), pf))
    with ThreadPoolExecutor(threads) as pool:
        oks = list(pool.map(band, range(threads)))
    return all(oks), out


def main():
    sections = set(sys.argv[1:]) or {"fixtures", "vectors", "digests", "digests_c", "digests_pf", "bands", "bands_all", "clipped"}
    os.makedirs(G, exist_ok=True)
    ref = ol.load_ref(); orc = ol.Oracle()
    orc.lib.orc_fnv1a64.restype = ctypes.c_uint64
    orc.lib.orc_fnv1a64.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    if "fixtures" in sections:
        make_fixtures(ref)
    if "vectors" in sections:
        make_vectors(ref)
    dpath = os.path.join(G, "digests_8192.json")
    W = H = 8192
    if "digests" in sections or not os.path.exists(dpath):
        dg = {"generator": "splitmix64, seed 0xD37E5000+k (tests/oracle_lib.py stream_u)", "width": 8192, "height": 8192,
              "reference_build": open(os.path.join(ROOT, "oracle/_ref/BUILD_INFO.txt")).read().strip(), "streams": {}}
        for f in F.FORMATS:
            for kind in ("U", "M"):
                if kind == "M" and f.name not in ("BPTC", "BPTC_FLOAT", "BPTC_SIGNED_FLOAT"): continue
                data = ol.stream_u(f, (W // 4) * (H // 4))
                if kind == "M": data = streams.stream_m(f, data)
                t = time.time(); ok, out = ref_linear_mt(ref, f, data, W, H); dt = time.time() - t
                fnv = orc.lib.orc_fnv1a64(out.ctypes.data, out.size)
                dg["streams"]["%s/%s" % (f.name, kind)] = {"ok": ok, "sha256": sha(out), "fnv1a64": "%016x" % fnv, "in_sha256": sha(data)}
                print(f.name, kind, ok, "%016x" % fnv, "%.2fs" % dt, flush=True)
    else:
        dg = json.load(open(dpath))
    if "digests_m_signed" in sections and "BPTC_SIGNED_FLOAT/M" not in dg["streams"]:
        # (round 6: stream M of the signed BC6H format added to an existing file without regenerating the other 40 digests)
        f = F.BY_NAME["BPTC_SIGNED_FLOAT"]
        data = streams.stream_m(f, ol.stream_u(f, (W // 4) * (H // 4)))
        ok, out = ref_linear_mt(ref, f, data, W, H)
        fnv = orc.lib.orc_fnv1a64(out.ctypes.data, out.size)
        dg["streams"]["BPTC_SIGNED_FLOAT/M"] = {"ok": ok, "sha256": sha(out), "fnv1a64": "%016x" % fnv, "in_sha256": sha(data)}
        print(f.name, "M", ok, "%016x" % fnv, flush=True)
    if "digests_c" in sections:
        # stream C: the bundled 64x64 fixture tiled over 8192^2 (tests/streams.py stream_c)
        for f in F.FORMATS:
            data = streams.stream_c(f, W // 4, H // 4)
            if data is None: continue
            ok, out = ref_linear_mt(ref, f, data, W, H)
            dg["streams"]["%s/C" % f.name] = {"ok": ok, "sha256": sha(out), "in_sha256": sha(data)}
            print(f.name, "C", ok, flush=True)
    if "digests_pf" in sections:
        # converted targets at full size (in-kernel epilogues)
        for key in [k for k in dg["streams"] if "/pf" in k]:
            del dg["streams"][key]
        for name, pf in (("BC1", F.PIXEL_FORMAT_BGRA8), ("BC1", F.PIXEL_FORMAT_RGB8), ("BC3", F.PIXEL_FORMAT_RGB8),
                         ("BPTC_FLOAT", F.PIXEL_FORMAT_FLOAT_BGRX16), ("BPTC_FLOAT", F.PIXEL_FORMAT_BGRX8), ("RGTC1", F.PIXEL_FORMAT_BGRX8),
                         ("EAC_RG11", F.PIXEL_FORMAT_RGB8), ("SIGNED_RGTC2", F.PIXEL_FORMAT_RGBA8), ("EAC_SIGNED_R11", F.PIXEL_FORMAT_BGRX8)):
            f = F.BY_NAME[name]
            data = ol.stream_u(f, (W // 4) * (H // 4))
            ok, out = ref_linear_mt(ref, f, data, W, H, pf)
            dg["streams"]["%s/U/pf%04X" % (name, pf)] = {"ok": ok, "sha256": sha(out), "in_sha256": sha(data), "bytes": int(out.size)}
            print(name, "U pf%04X" % pf, ok, flush=True)
    if "bands" in sections:
        # one GPU's band of the sharded 32768^2 configs (BASELINE configs[4], north_star): sha256 of the WHOLE band
        dg.setdefault("bands", {})
        for name, bw, bh in (("BC1", 32768, 8192), ("BPTC_FLOAT", 32768, 4096)):
            f = F.BY_NAME[name]
            data = ol.stream_u(f, (bw // 4) * (bh // 4))
            t = time.time(); ok, out = ref_linear_mt(ref, f, data, bw, bh); dt = time.time() - t
            dg["bands"]["%s/%dx%d" % (name, bw, bh)] = {"ok": ok, "sha256": sha(out), "in_sha256": sha(data), "bytes": int(out.size)}
            print(name, bw, bh, ok, "%.1fs" % dt, flush=True)
    if "bands_all" in sections:
        # EVERY band of the two sharded 32768^2 images (BASELINE configs[4] / north_star: BC1 over 4 GPUs, BC6H over 8), generated with
        # the GLOBAL stream offsets a rank uses (sharding.shard_of -> ol.stream_u_slice): band g of G is what rank g of G decodes
        from detex_amd import sharding
        dg["bands_all"] = {}
        # (BC1 also in eighths: a rank of bench.py --gpus N, N in 1, 2, 4, 8, checks the 8 / N eighths its band consists of)
        for name, side, world in (("BC1", 32768, 4), ("BC1", 32768, 8), ("BPTC_FLOAT", 32768, 8)):
            f = F.BY_NAME[name]
            for g in range(world):
                sh = sharding.shard_of(g, world, f, side, side)
                data = ol.stream_u_slice(f, sh.row0 * (side // 4), (sh.row1 - sh.row0) * (side // 4))
                t = time.time(); ok, out = ref_linear_mt(ref, f, data, side, sh.px_rows); dt = time.time() - t
                assert out.size == sh.out_bytes
                dg["bands_all"]["%s/%d/%dof%d" % (name, side, g, world)] = {"ok": ok, "sha256": sha(out), "in_sha256": sha(data), "bytes": int(out.size),
                                                                           "row0": sh.row0, "row1": sh.row1}
                print("band", name, g, world, ok, "%.1fs" % dt, flush=True)
                del out
    if "clipped" in sections:
        # large textures with clipped last block column / row (texture.c:116-120, 132-136): interior through the throughput
        # kernel (aligned and dword-aligned rows), edge strips pixel by pixel, and widths whose rows are not even dword-aligned
        dg["clipped"] = {}
        for name, w, h, pf in CLIPPED_CASES:
            f = F.BY_NAME[name]
            wb, hb = (w + 3) // 4, (h + 3) // 4
            data = ol.stream_u(f, wb * hb)
            ok, out = ref_linear_clipped_mt(ref, f, data, w, h, pf)
            key = "%s/%dx%d" % (name, w, h) + ("/pf%04X" % pf if pf else "")
            dg["clipped"][key] = {"ok": ok, "sha256": sha(out), "in_sha256": sha(data), "bytes": int(out.size)}
            print("clipped", key, ok, flush=True)
    json.dump(dg, open(dpath, "w"), indent=1, sort_keys=True)


CLIPPED_CASES = [("BC1", 8190, 8190, None), ("BPTC", 4093, 4091, None), ("BPTC_FLOAT", 4093, 2047, None), ("RGTC1", 4093, 2047, None),
                 ("RGTC2", 4094, 2046, None), ("EAC_R11", 4094, 2045, None), ("BC3", 4094, 2046, F.PIXEL_FORMAT_RGB8),
                 ("ETC2_EAC", 8190, 4096, None), ("BC1", 8192, 8190, None)]


def ref_linear_clipped_mt(ref, f, data, W, H, pf=None, threads=None):
    """as ref_linear_mt for a texture whose width / height need not be multiples of four"""
    from concurrent.futures import ThreadPoolExecutor
    threads = threads or min(16, os.cpu_count() or 1)
    pf = (f.texture_format & 0xFFFF) if pf is None else pf
    px = 1 + ((pf & 0xF00) >> 8)
    wb, hb = (W + 3) // 4, (H + 3) // 4
    out = np.empty(W * H * px, np.uint8)
    data = np.ascontiguousarray(data)

    def band(g):
        r0, r1 = g * hb // threads, (g + 1) * hb // threads
        if r1 <= r0:
            return True
        rows = min(r1 * 4, H) - r0 * 4
        tex = ol.DetexTexture(f.texture_format, ol._ptr(data[r0 * wb * f.block_bytes:]), W, rows, wb, r1 - r0)
        return bool(ref.lib.detexDecompressTextureLinear(ctypes.byref(tex), ol._ptr(out[r0 * 4 * W * px:]), pf))
    with ThreadPoolExecutor(threads) as pool:
        oks = list(pool.map(band, range(threads)))
    return all(oks), out


def make_fixtures(ref):
    fx = {}
    for f in F.FORMATS:
        if not f.fixture: continue
        shutil.copyfile(os.path.join(REF_DIR, f.fixture), os.path.join(G, f.fixture))
        k = read_ktx(os.path.join(G, f.fixture))
        assert k["format"] is f and (k["width"], k["height"]) == (64, 64)
        ent = {}
        for pf in F.accepted_pixel_formats(f):
            ok, out = ref.linear(f, k["data"], 64, 64, pixel_format=pf)
            ent["0x%04X" % pf] = {"ok": ok, "sha256": sha(out), "bytes": int(out.size)}
        # an 8-bit RGB(A) target this library refuses must be one the reference cannot reach either
        for pf in F.ALL_TARGETS[:5]:
            if pf not in F.accepted_pixel_formats(f):
                ok, _ = ref.linear(f, k["data"], 64, 64, pixel_format=pf)
                assert not ok, (f.name, hex(pf), "the reference converts this; the library should offer it")
        fx[f.name] = ent
    json.dump(fx, open(os.path.join(G, "fixtures.json"), "w"), indent=1, sort_keys=True)


def make_vectors(ref):
    vec, clip, mf = {}, {}, {}
    for f in F.FORMATS:
        blocks, labels = streams.forced_stream(f)
        n = len(blocks)
        fn = ref.block_fn(f)
        out = np.zeros((n, 16 * f.pixel_bytes), np.uint8); okv = np.zeros(n, np.uint8)
        for i in range(n):
            okv[i] = fn(ol._ptr(blocks[i]), 0xFFFFFFFF, 0, ol._ptr(out[i]))
        out[okv == 0] = 0
        vec[f.name + "/in"] = blocks; vec[f.name + "/out"] = out; vec[f.name + "/ok"] = okv
        ent = {}
        for mask, flags in streams.MASK_FLAG_MATRIX:
            o2 = np.zeros_like(out); ok2 = np.zeros(n, np.uint8)
            for i in range(n):
                ok2[i] = fn(ol._ptr(blocks[i]), mask, flags, ol._ptr(o2[i]))
            o2[ok2 == 0] = 0
            ent["%08X/%X" % (mask, flags)] = {"ok_bits": np.packbits(ok2).tobytes().hex(), "sha256": sha(o2)}
        mf[f.name] = ent
        flat = blocks.reshape(-1)
        for (w, h) in streams.CLIP_SIZES:
            wb, hb = (w + 3) // 4, (h + 3) // 4
            need = wb * hb * f.block_bytes
            data = np.resize(flat, need)
            ok, o = ref.linear(f, data, w, h)
            clip["%s/%dx%d" % (f.name, w, h)] = o
            clip["%s/%dx%d/ok" % (f.name, w, h)] = np.array([ok])
            if (w, h) in streams.CLIP_SIZES_CONVERTED:
                for pf in F.accepted_pixel_formats(f):
                    if F.epilogue_kind(f, pf):
                        ok2, o2 = ref.linear(f, data, w, h, pixel_format=pf)
                        assert ok2 == ok
                        clip["%s/%dx%d/pf%04X" % (f.name, w, h, pf)] = o2
        print("vectors", f.name, n, "blocks", int(okv.sum()), "valid", flush=True)
    np.savez_compressed(os.path.join(G, "forced_vectors.npz"), **vec)
    np.savez_compressed(os.path.join(G, "clip.npz"), **clip)
    json.dump(mf, open(os.path.join(G, "maskflags.json"), "w"), indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
