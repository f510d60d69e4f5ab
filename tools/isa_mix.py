#!/usr/bin/env python3
"""Static instruction mix of the library's gfx950 kernels (container; no GPU).
   python tools/isa_mix.py 'decode_linearINS_13DecBPTCFloatTILb0ELb0EEELi0ELb1E' [--dump]   one kernel (substring of the mangled name): counts,
                                                                                             the thirty most frequent opcodes, optionally the listing
   python tools/isa_mix.py --all ['decode_linearI.*Lb1E+v']                                  one line per kernel whose name matches the regex
Compiles the library's device translation units to assembly once (build/scratch/isa_*.s, reused while newer than the sources) and counts the instructions between
a kernel's label and its s_endpgm.  The decoders are (nearly) branch-free, so the static VALU count is what a wave executes; kernels with
wave-uniform alternatives (BC7's per-record copies, ETC2's paths) count every alternative.
The issue-cycle estimate prices each VALU instruction with the per-class rates measured by tools/ubench/valu_rates.hip (profiles/r0*/
valu_rates.txt): adds, subs, logic ops, right shifts, moves and v_bitop3 on VGPR / inline operands ~2.5 cycles per wave64 instruction,
the same ops with an SGPR source 4.2, everything else (multiplies, bit-field ops, v_perm, left shifts, packed and SDWA forms) ~4.4."""
import collections, glob, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
srcs = [p for ext in ("*.h", "*.hip", "*.inc") for p in glob.glob(os.path.join(ROOT, "detex_amd", "csrc", ext))]
newest = max(os.path.getmtime(p) for p in srcs)
lines, jobs = [], []
for unit in sorted(glob.glob(os.path.join(ROOT, "detex_amd", "csrc", "*.hip"))):
    asm = os.path.join(ROOT, "build", "scratch", "isa_%s.s" % os.path.splitext(os.path.basename(unit))[0])
    if not os.path.exists(asm) or os.path.getmtime(asm) < newest:
        os.makedirs(os.path.dirname(asm), exist_ok=True)
        jobs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-Wno-pass-failed",
                                      "-o", asm, unit] + os.environ.get("ISA_MIX_FLAGS", "").split(), stderr=subprocess.DEVNULL))
    lines.append(asm)
assert all(j.wait() == 0 for j in jobs), "hipcc failed"
lines = [l for asm in lines for l in open(asm).read().splitlines()]
FAST = ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b32", "v_bitop3_b32")


def cost(line):
    op = line.split()[0]
    if not op.startswith("v_"):
        return 0.0
    base = op.replace("_e32", "").replace("_e64", "")
    if base in FAST:
        return 4.2 if base != "v_mov_b32" and re.search(r"\bs\d+|\bs\[|\bvcc", line[len(op):]) else 2.5
    return 4.4


def kernels():
    for i, l in enumerate(lines):
        if l.startswith("_ZN") and l.split(";")[0].rstrip().endswith(":"):
            end = next((k for k in range(i + 1, len(lines)) if lines[k].strip().startswith("s_endpgm") or lines[k].startswith("_ZN")), None)
            if end is None or not lines[end].strip().startswith("s_endpgm"):
                continue				# a data symbol
            yield l.split(":")[0], [x.strip() for x in lines[i + 1:end] if x.startswith("\t") and not x.strip().startswith((".", ";"))]


def vgprs(name):
    i = next((k for k, l in enumerate(lines) if l.strip() == ".amdhsa_kernel " + name), None)
    if i is None:
        return -1
    return int(next(l for l in lines[i:i + 60] if "next_free_vgpr" in l).split()[-1])


if "--all" in sys.argv:
    rest = [a for a in sys.argv[1:] if a != "--all"]
    pat = re.compile(rest[0] if rest else r"decode_linearI.*Lb1E+v")
    for name, body in kernels():
        if pat.search(name):
            ops = collections.Counter(l.split()[0] for l in body)
            cls = lambda p: sum(c for o, c in ops.items() if o.startswith(p))
            print("%-70s VALU %5d  SALU %4d  LDS %3d  global %2d  branches %2d  VGPRs %3d  est. cycles %5.0f" % (
                re.sub(r"^_ZN8detexhip\d+", "", name)[:70], cls("v_"), cls("s_") - ops.get("s_waitcnt", 0), cls("ds_"), cls("global_"), cls("s_cbranch"),
                vgprs(name), sum(cost(l) for l in body)))
    sys.exit(0)

pat = sys.argv[1]
name, body = next((n, b) for n, b in kernels() if pat in n)
ops = collections.Counter(l.split()[0] for l in body)
cls = lambda p: sum(c for o, c in ops.items() if o.startswith(p))
print(name)
print("instructions %d: VALU %d (cndmask %d, sdwa %d, bitop3 %d, perm %d, pk %d)  SALU %d  LDS %d  global %d  s_waitcnt %d  VGPRs %d" % (
    len(body), cls("v_"), sum(c for o, c in ops.items() if "cndmask" in o), sum(c for o, c in ops.items() if o.endswith("_sdwa")),
    ops.get("v_bitop3_b32", 0), ops.get("v_perm_b32", 0), cls("v_pk_"), cls("s_") - ops.get("s_waitcnt", 0), cls("ds_"), cls("global_"), ops.get("s_waitcnt", 0), vgprs(name)))
print("  estimated VALU issue cycles per wave %.0f (fast-class instructions %d)" % (sum(cost(l) for l in body), sum(1 for l in body if cost(l) == 2.5)))
print("  " + "  ".join("%s %d" % kv for kv in sorted(ops.items(), key=lambda x: -x[1])[:30]))
if "--dump" in sys.argv:
    print("\n".join(body))
