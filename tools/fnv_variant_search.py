#!/usr/bin/env python3
"""SURVEY.md Appendix B quotes FNV-1a-64 values for the reference's output on the 8192^2 stream-U inputs (BC1 `fda3ff8943e5de3c` ...).
The committed digests (tests/golden/digests_8192.json, e.g. BC1/U fnv1a64 `a675b88e38cab606`) come from the COMPILED REFERENCE
(oracle/_ref) on tests/oracle_lib.py's splitmix64 stream, which follows the survey's recipe to the letter -- and do not equal the
survey's values.  This script is the search for a generator / digest variant that would: build container only (needs oracle/_ref).

Variants tried (round 5), BC1 8192^2, none reproduces fda3ff8943e5de3c:
  generator   first output = mix(seed + 1*gamma) [the survey's text: `state += gamma` first] | mix(seed + 0*gamma) | mix(seed + 2*gamma);
              seeds 0xD37E5000, 0xD37E5001, 0xD37E4FFF, 0;  words stored little-endian | big-endian;  low / high 32-bit halves as u32
              words (two draws per block);  one byte per draw (low byte | high byte)
  digest      FNV-1a over bytes | FNV-1 (multiply first) | FNV-1a over u32 words | over u64 words | bytes sign-extended (a `char` bug) |
              32-bit prime 0x01000193 in 64-bit state;  over the linear image (RGBX8 and RGBA8 targets: same bytes) | the block-major
              (tiled) image | the INPUT stream
The survey's generator program was not committed, so its values cannot be re-derived; parity does not depend on them: the fixture
sha256 of Appendix B (17/17) DO match, and every committed digest is regenerated from the compiled reference by tools/make_goldens.py.
"""
import ctypes
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol                     # noqa: E402
from detex_amd import formats as F          # noqa: E402

TARGET = 0xfda3ff8943e5de3c
C_SRC = r'''
#include <stdint.h>
#include <stddef.h>
#define LOOP(T, step) uint64_t h = 0xcbf29ce484222325ull; const T *q = (const T *)p; for (size_t i = 0; i < n / sizeof(T); i++) { step } return h;
uint64_t d_fnv1a(const void *p, size_t n)   { LOOP(uint8_t,  h ^= q[i]; h *= 0x100000001b3ull;) }
uint64_t d_fnv1(const void *p, size_t n)    { LOOP(uint8_t,  h *= 0x100000001b3ull; h ^= q[i];) }
uint64_t d_w32(const void *p, size_t n)     { LOOP(uint32_t, h ^= q[i]; h *= 0x100000001b3ull;) }
uint64_t d_w64(const void *p, size_t n)     { LOOP(uint64_t, h ^= q[i]; h *= 0x100000001b3ull;) }
uint64_t d_signed(const void *p, size_t n)  { LOOP(int8_t,   h ^= (uint64_t)(int64_t)q[i]; h *= 0x100000001b3ull;) }
uint64_t d_prime32(const void *p, size_t n) { LOOP(uint8_t,  h ^= q[i]; h *= 0x01000193ull;) }
'''
DIGESTS = ("d_fnv1a", "d_fnv1", "d_w32", "d_w64", "d_signed", "d_prime32")
M64 = np.uint64


def draws(seed, n, first):
    with np.errstate(over="ignore"):
        z = M64(seed) + M64(0x9E3779B97F4A7C15) * np.arange(first, first + n, dtype=np.uint64)
        z = (z ^ (z >> M64(30))) * M64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> M64(27))) * M64(0x94D049BB133111EB)
        return z ^ (z >> M64(31))


def main():
    if not ol.have_ref():
        sys.exit("needs oracle/_ref (build container)")
    tmp = tempfile.mkdtemp()
    open(os.path.join(tmp, "d.c"), "w").write(C_SRC)
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", os.path.join(tmp, "d.so"), os.path.join(tmp, "d.c")])
    lib = ctypes.CDLL(os.path.join(tmp, "d.so"))
    for d in DIGESTS:
        getattr(lib, d).restype = ctypes.c_uint64
        getattr(lib, d).argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    ref = ol.load_ref()
    fmt, side = F.BY_NAME["BC1"], 8192
    n = (side // 4) ** 2
    tried = hits = 0

    def check(label, buf):
        nonlocal tried, hits
        for d in DIGESTS:
            tried += 1
            h = getattr(lib, d)(buf.ctypes.data, buf.size)
            if h == TARGET:
                hits += 1
                print("MATCH:", label, d)

    for seed in (0xD37E5000, 0xD37E5001, 0xD37E4FFF, 0):
        for first in (1, 0, 2):
            w = draws(seed, n, first)
            inputs = {"LE": w.view(np.uint8), "BE": w.byteswap().view(np.uint8)}
            if seed == 0xD37E5000 and first < 2:
                w2, w8 = draws(seed, 2 * n, first), draws(seed, 8 * n, first)
                inputs.update({"lo32": (w2 & M64(0xFFFFFFFF)).astype(np.uint32).view(np.uint8), "hi32": (w2 >> M64(32)).astype(np.uint32).view(np.uint8),
                               "lobyte": (w8 & M64(0xFF)).astype(np.uint8), "hibyte": (w8 >> M64(56)).astype(np.uint8)})
            for name, data in inputs.items():
                data = np.ascontiguousarray(data)
                label = "seed %#x first draw %d %s" % (seed, first, name)
                check(label + " linear", ref.linear(fmt, data, side, side)[1])
                if seed == 0xD37E5000 and name == "LE":
                    check(label + " tiled", ref.tiled(fmt, data, side // 4, side // 4)[1])
                    check(label + " input", data)
    print("%d (generator, digest) variants tried, %d reproduce %016x" % (tried, hits, TARGET))


if __name__ == "__main__":
    main()
