#!/usr/bin/env python3
"""Compare the gfx950 kernels of two builds instruction by instruction (container; no GPU): refactorings of the host side or of the
header structure must leave every kernel's instruction stream unchanged.
   python tools/isa_diff.py old.s new_a.s [new_b.s ...]     (device assembly: hipcc -S --cuda-device-only)
Kernels are matched by mangled name; label definitions and comments are ignored, branch targets compared by their number inside the kernel, symbol addresses (s_getpc / s_add of table symbols) are kept."""
import re, sys


def kernels(path):
    lines = open(path).read().splitlines()
    out = {}
    i = 0
    while i < len(lines):
        l = lines[i]
        if l.startswith("_ZN") and l.split(";")[0].rstrip().endswith(":"):
            name = l.split(":")[0]
            body = []
            k = i + 1
            while k < len(lines) and not lines[k].strip().startswith("s_endpgm") and not lines[k].startswith("_ZN"):
                t = lines[k].strip()
                if lines[k].startswith("\t") and not t.startswith((".", ";")):
                    body.append(re.sub(r"\.LBB\d+_", ".LBB_", re.sub(r"\s*;.*$", "", t)))     # (branch labels carry the function's number in its translation unit)
                k += 1
            if k < len(lines) and lines[k].strip().startswith("s_endpgm"):
                out[name] = body
            i = k
        i += 1
    return out


old = kernels(sys.argv[1])
new = {}
for p in sys.argv[2:]:
    new.update(kernels(p))
same = differ = 0
for name in sorted(set(old) | set(new)):
    if name not in new:
        print("ONLY IN OLD", name)
    elif name not in old:
        print("ONLY IN NEW", name)
    elif old[name] != new[name]:
        differ += 1
        print("DIFFERS (%d -> %d instructions)" % (len(old[name]), len(new[name])), name)
    else:
        same += 1
print("%d kernels identical, %d differ, %d old, %d new" % (same, differ, len(old), len(new)))
