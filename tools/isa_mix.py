#!/usr/bin/env python3
"""Static instruction mix of one kernel of the library (container; no GPU):
   python tools/isa_mix.py 'decode_linearINS_13DecBPTCFloatTILb0ELb0EEELi0ELb1E' [--dump]
Compiles detexhip.hip to assembly (build/scratch/detexhip.s; reused if newer than the sources) and counts the
instructions between the kernel's label and its s_endpgm.  Branch-free decoders: the static VALU count is what a wave executes."""
import collections, glob, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
asm = os.path.join(ROOT, "build", "scratch", "detexhip.s")
srcs = glob.glob(os.path.join(ROOT, "detex_amd", "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "detex_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "detex_amd", "csrc", "*.inc"))
if not os.path.exists(asm) or os.path.getmtime(asm) < max(os.path.getmtime(p) for p in srcs):
    os.makedirs(os.path.dirname(asm), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-Wno-pass-failed",
                           "-o", asm, os.path.join(ROOT, "detex_amd", "csrc", "detexhip.hip")], stderr=subprocess.DEVNULL)
pat = sys.argv[1]
lines = open(asm).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and pat in l.split(":")[0] and l.rstrip().split(";")[0].rstrip().endswith(":"))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = [l.strip() for l in lines[start + 1:end] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
ops = collections.Counter(l.split()[0] for l in body)
cls = lambda p: sum(c for o, c in ops.items() if o.startswith(p))
print(lines[start].split(":")[0])
print("instructions %d: VALU %d (cndmask %d, sdwa %d, bitop3 %d, perm %d, pk %d)  SALU %d  LDS %d  global %d  s_waitcnt %d" % (
    len(body), cls("v_"), sum(c for o, c in ops.items() if "cndmask" in o), sum(c for o, c in ops.items() if o.endswith("_sdwa")),
    ops.get("v_bitop3_b32", 0), ops.get("v_perm_b32", 0), cls("v_pk_"), cls("s_") - ops.get("s_waitcnt", 0), cls("ds_"), cls("global_"), ops.get("s_waitcnt", 0)))
# issue cost per wave64 instruction from tools/ubench/valu_rates.hip (profiles/r0*/valu_rates.txt): adds, subs, logic ops, right shifts,
# moves and v_bitop3 with VGPR / inline operands ~2.5 cycles; the same with an SGPR source, and everything else, ~4.4
FAST = ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b32", "v_bitop3_b32")
def cost(line):
    op = line.split()[0]
    if not op.startswith("v_"):
        return 0.0
    base = op.replace("_e32", "").replace("_e64", "")
    if base in FAST and "_sdwa" not in op and "_dpp" not in op:
        operands = line[len(op):]
        return 4.2 if re.search(r"\bs\d+|\bs\[|vcc|0x[0-9a-f]{0}(?=$)", operands) and base != "v_mov_b32" else 2.5
    return 4.4
cycles = sum(cost(l) for l in body)
print("  estimated VALU issue cycles per wave %.0f (fast-class ops %d)" % (cycles, sum(1 for l in body if cost(l) == 2.5)))
print("  " + "  ".join("%s %d" % kv for kv in sorted(ops.items(), key=lambda x: -x[1])[:30]))
if "--dump" in sys.argv:
    print("\n".join(body))
