#!/usr/bin/env python3
"""Static instruction mix of the gfx950 kernels (compile to ISA with hipcc -S, count per kernel)."""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/detexhip_all.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                       "-o", out, os.path.join(ROOT, "detex_amd/csrc/detexhip.hip")], stderr=subprocess.DEVNULL)
txt = open(out).read()
pat = sys.argv[1] if len(sys.argv) > 1 else r"decode_linearI.*Lb1E+v"
for f in re.split(r"\n\t\.globl\t", txt)[1:]:
    name = f.split("\n", 1)[0].strip()
    if not re.search(pat, name): continue
    body = f.split(".Lfunc_end")[0]
    ins = [l.strip().split()[0] for l in body.split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    valu = sum(i.startswith("v_") for i in ins); salu = sum(i.startswith("s_") for i in ins)
    mem = sum(i.startswith(("global_", "flat_", "buffer_", "ds_", "scratch_")) for i in ins)
    slow = sum(i.startswith(("v_mul_lo", "v_mul_hi", "v_lshrrev_b64", "v_lshlrev_b64", "v_mad_u64", "v_ashrrev_i64")) for i in ins)
    br = sum(i.startswith("s_cbranch") for i in ins)
    # issue-cost estimate from tools/ubench (cycles per wave64 instruction per SIMD at 8 waves/SIMD)
    FAST = ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_mov_b32", "v_lshrrev_b32",
            "v_ashrrev_i32", "v_bitop3_b32", "v_add_co", "v_sub_co")
    cyc = sum(2.5 if i.startswith(FAST) and not i.endswith("_sdwa") else 4.4 for i in ins if i.startswith("v_"))
    m = re.search(r"\.vgpr_count:\s+(\d+)", f)
    short = re.sub(r"^_ZN8detexhip\d+", "", name)[:60]
    print("%-62s valu=%5d salu=%4d mem=%3d slow=%3d br=%3d est_cycles=%5d" % (short, valu, salu, mem, slow, br, cyc))
