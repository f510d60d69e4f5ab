"""Reference points for the HBM roofline on this box, and the data-dependence question of DESIGN.md section 8 / profiles/AB_RECORD.md:
write-only fills with the decode kernels' store shape (tools/ubench/hbm_ref.hip) for seven data patterns at sizes on both
sides of the 256 MiB Infinity Cache, non-temporal and ordinary stores, plus 16-byte-vector copies.
usage: python tools/gpu_hbm_ref.py [out.jsonl]      (one JSON line per measurement, a table on stderr)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import hbmref

out = open(sys.argv[1], "w") if len(sys.argv) > 1 else None
def emit(row):
    line = json.dumps(row)
    print(line, flush=True)
    if out:
        out.write(line + "\n"); out.flush()

for mib in (128, 256, 512, 1024, 4096):
    nbytes = mib << 20
    buf = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    launches = 60 if mib <= 512 else (30 if mib <= 1024 else 10)
    for nt in (True, False):
        for pat in sorted(hbmref.PATTERNS):
            if not nt and pat not in (0, 2):
                continue
            gbps, us = hbmref.fill_GBps(nbytes, pat, nt, launches, buf)
            emit({"op": "fill", "MiB": mib, "pattern": pat, "pattern_name": hbmref.PATTERNS[pat], "nontemporal": nt, "us": round(us, 1), "TBps_written": round(gbps / 1e3, 3)})
    del buf
    torch.cuda.empty_cache()
for mib in (256, 1024):
    for nt in (True, False):
        gbps, us = hbmref.copy_GBps(mib << 20, nt, 30)
        emit({"op": "copy", "MiB": mib, "nontemporal": nt, "us": round(us, 1), "TBps_read_plus_written": round(gbps / 1e3, 3)})
# how many 16-byte stores a wave issues: 1 (a workgroup writes 4 KiB -- the shape of torch's fill kernel) .. 8
for mib in (256, 1024):
    for rows in (1, 2, 4, 8):
        for nt in (True, False, 2):
            gbps, us = hbmref.fill_rows_GBps(mib << 20, rows, nt, 30)
            emit({"op": "fill_rows", "MiB": mib, "stores_per_lane": rows, "nontemporal": bool(nt), "wave_contiguous": nt == 2, "us": round(us, 1), "TBps_written": round(gbps / 1e3, 3)})
for rows in (1, 4):
    gbps, us = hbmref.fill_rows_GBps(1 << 30, rows, 3, 30)
    emit({"op": "fill_rows_one_wave_workgroups", "MiB": 1024, "stores_per_lane": rows, "us": round(us, 1), "TBps_written": round(gbps / 1e3, 3)})
# the fill in the decode kernels' image layout, and through stores that are only dword-aligned
for (wbytes, h) in ((32768, 8192), (65536, 8192), (65536, 16384)):
    for pat in (0, 2):
        gbps, us = hbmref.fill_image_GBps(wbytes, h, pat, 30)
        emit({"op": "fill_image", "width_bytes": wbytes, "height": h, "MiB": wbytes * h >> 20, "pattern": pat, "us": round(us, 1), "TBps_written": round(gbps / 1e3, 3)})
# ... with the resident workgroups per CU capped (unused dynamic LDS), as the decode kernels' launches do
for wg in (0, 7, 6, 5, 4, 3, 2):
    for (wbytes, h) in ((32768, 8192), (65536, 16384)):
        gbps, us = hbmref.fill_image_GBps(wbytes, h, 2, 40, 0, wg)
        emit({"op": "fill_image_resident", "workgroups_per_cu": wg, "width_bytes": wbytes, "height": h, "MiB": wbytes * h >> 20, "us": round(us, 1), "TBps_written": round(gbps / 1e3, 3)})
# ... with padded row pitches: what the memory system makes of an image whose rows are not a power of two apart
for pad in (0, 64, 128, 256, 4096 + 64, 16, 32, 48):
    gbps, us = hbmref.fill_image_GBps(32768, 8192, 2, 60, 32768 + pad)
    emit({"op": "fill_image_pitch", "width_bytes": 32768, "height": 8192, "pitch": 32768 + pad, "pitch_mod_64": pad % 64, "us": round(us, 1), "TBps_written": round(gbps / 1e3, 3)})
for off in (0, 4, 8, 12, 16, 32, 48, 64, 112):
    gbps, us = hbmref.fill_unaligned_GBps(256 << 20, off, 30)
    emit({"op": "fill_dword_aligned_stores", "MiB": 256, "offset": off, "us": round(us, 1), "TBps_written": round(gbps / 1e3, 3)})
# torch's own fill / zero / copy for comparison with round 1's table
for mib in (256, 1024):
    n = (mib << 20) // 4
    a = torch.empty(n, dtype=torch.int32, device="cuda"); b = torch.empty_like(a)
    for name, fn, mult in (("torch fill_(7)", lambda: a.fill_(7), 1), ("torch zero_()", lambda: a.zero_(), 1), ("torch copy_", lambda: b.copy_(a), 2)):
        us = hbmref.time_us(fn, 30)
        emit({"op": name, "MiB": mib, "us": round(us, 1), "TBps": round(mult * (mib << 20) / (us * 1e-6) / 1e12, 3)})
    del a, b
