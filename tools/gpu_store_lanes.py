#!/usr/bin/env python3
"""Does the write rate depend on how a tile's texel rows are dealt out?  Image-layout fills (tools/ubench/hbm_ref.hip): the decode kernels'
shape (every lane four stores, one per texel row: fill_image) against ONE store per lane -- wave w of the workgroup writes texel row w of 64
blocks (shape 0), a workgroup writes 4 KiB of one image row (shape 1), two stores per lane (shape 2) -- non-temporal and ordinary stores;
the four-stores shape in 64- / 128- / 512- / 1024-lane workgroups and with the rows rotated per wave; the one-store shape on persistent grids
(no sync / a barrier / the store acknowledged per tile); four stores each acknowledged before the next -- at the headline's image (32 KiB rows
x 8192) and 64 KiB x 16384.  usage: python tools/gpu_store_lanes.py [rounds]      GPU box."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import hbmref

lib = hbmref.load()
lib.hbmref_fill_image_lane.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p]
lib.hbmref_fill_image_group.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p]
lib.hbmref_fill_image_persistent.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p]
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for wb, h in ((32768, 8192), (65536, 16384)):
    buf = torch.empty(wb * h, dtype=torch.uint8, device="cuda")
    seed = [0]

    def image4():
        seed[0] += 977
        assert lib.hbmref_fill_image(buf.data_ptr(), wb, h, 0, 2, seed[0], st) == 0

    def lane(shape, nt):
        def f():
            seed[0] += 977
            assert lib.hbmref_fill_image_lane(buf.data_ptr(), wb, h, 0, shape, nt, seed[0], st) == 0
        return f
    cases = [("four stores per lane, nt (the decode kernels' shape)", image4)]
    for shape, name in ((0, "wave = texel row, 1 KiB x 4 rows per workgroup"), (1, "workgroup = 4 KiB of one row"), (2, "wave = texel row, two stores per lane")):
        for nt in (1, 0):
            cases.append(("%s, %s" % (name, "nt" if nt else "ordinary"), lane(shape, nt)))
    def group(lanes, rotate):
        def f():
            seed[0] += 977
            assert lib.hbmref_fill_image_group(buf.data_ptr(), wb, h, lanes, rotate, seed[0], st) == 0
        return f
    cases += [("four stores per lane, 64-lane workgroups, nt", group(64, 0)), ("four stores per lane, 128-lane workgroups, nt", group(128, 0)),
              ("four stores per lane, 256 lanes, wave w starts at row w, nt", group(256, 1)), ("four stores per lane, 256 lanes (group kernel), nt", group(256, 0)),
              ("four stores per lane, 512-lane workgroups, nt", group(512, 0)), ("four stores per lane, 1024-lane workgroups, nt", group(1024, 0))]
    def persistent(sync, groups):
        def f():
            seed[0] += 977
            assert lib.hbmref_fill_image_persistent(buf.data_ptr(), wb, h, sync, groups, seed[0], st) == 0
        return f
    for groups in (2048, 4096):
        for sync, name in ((0, "no sync"), (1, "barrier per tile"), (2, "store acknowledged before the next")):
            cases.append(("wave = texel row on a persistent grid of %d workgroups, %s, nt" % (groups, name), persistent(sync, groups)))
    cases.append(("four stores per lane, each acknowledged before the next, nt", persistent(4, 0)))
    for _ in range(200):
        image4()                      # settle
    for r in range(rounds):
        for name, fn in cases:
            us = hbmref.time_us(fn, 60, 10)
            print(json.dumps({"image": "%d x %d" % (wb, h), "MiB": wb * h >> 20, "round": r, "case": name, "us": round(us, 2), "TBps_written": round(wb * h / us / 1e6, 3)}), flush=True)
    del buf
    torch.cuda.empty_cache()
