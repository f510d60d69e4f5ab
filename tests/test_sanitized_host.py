"""Host-side code of the library under AddressSanitizer + UndefinedBehaviorSanitizer.

The KTX loader (detex_amd/csrc/ktx_loader.cpp; reference: ktx.c:36-176) is the one part of the library that parses file
content; it is compiled here with g++ and both sanitizers (tests/host_san/ktx_san_main.cpp, `make ktx-san`) and fed a corpus of
hostile files: every truncation of a valid mip chain, header fields replaced by extreme values, the byte-swapped form, key /
value sizes that point outside the file, and seeded random byte corruption.  Any sanitizer report fails the test.

CONTAINER ONLY.  The GPU pool runs no sanitizer builds: the instrumentation flags live in tests/host_san/san.mk, and that file, this
module and the instrumented binaries are listed in .gpurunignore.  The uninstrumented builds of the same programs run on the GPU box
(tests/test_gpu_host_programs.py); earlier in round 6, before the pool's rule, the instrumented ones ran there too
(profiles/r06/api_san_gpu.txt; the teardown program's three endings in that round's GPU test log)."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "tests", "host_san")
KTX_ID = bytes([0xAB, 0x4B, 0x54, 0x58, 0x20, 0x31, 0x31, 0xBB, 0x0D, 0x0A, 0x1A, 0x0A])


@pytest.fixture(scope="module")
def ktx_san():
    exe = os.path.join(SAN, "ktx_san")
    subprocess.check_call(["make", "-s", "-C", ROOT, "ktx-san"])
    return exe


def _ktx(levels, gl=0x83F0, swapped=False, kv=b"", declared_levels=None, w=None, h=None):
    """a KTX1 file with the given [(w, h, payload)] mip levels"""
    e = ">" if swapped else "<"
    w0, h0, _ = levels[0]
    hdr = KTX_ID + struct.pack(e + "13I", 0x04030201, 0, 1, 0, gl, 0x1907, w if w is not None else w0, h if h is not None else h0, 0, 0, 1,
                               len(levels) if declared_levels is None else declared_levels, len(kv))
    body = kv
    for i, (_, _, payload) in enumerate(levels):
        body += struct.pack(e + "I", len(payload)) + payload
        if i + 1 < len(levels):
            body += b"\0" * (3 - (len(payload) + 3) % 4)
    return hdr + body


def test_ktx_loader_survives_a_hostile_corpus(ktx_san, tmp_path):
    rng = np.random.default_rng(20260927)
    chain = [(16, 12, bytes(rng.integers(0, 256, 4 * 3 * 8, dtype=np.uint8))), (8, 6, bytes(rng.integers(0, 256, 2 * 2 * 8, dtype=np.uint8))),
             (4, 3, bytes(rng.integers(0, 256, 8, dtype=np.uint8)))]
    good = _ktx(chain, kv=b"\x10\0\0\0KTXorient\0S=r\0\0\0")
    files = {"good": good, "swapped": _ktx(chain, swapped=True), "bc7": _ktx([(8, 8, bytes(64))], gl=0x8E8C)}
    for n in range(0, len(good), 3):                                   # every truncation
        files["trunc%04d" % n] = good[:n]
    extremes = [0, 1, 3, 32768, 32769, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFF]
    for field in range(3, 16):                                        # each header word replaced by extreme values
        for v in extremes:
            b = bytearray(good)
            struct.pack_into("<I", b, 12 + 4 * (field - 3), v)
            files["hdr%02d_%08x" % (field, v)] = bytes(b)
    for v in extremes:                                                # ... and the first image-size word
        b = bytearray(good)
        struct.pack_into("<I", b, 64 + 20, v)
        files["size_%08x" % v] = bytes(b)
    files["levels_more_than_present"] = _ktx(chain, declared_levels=9)
    files["huge_dims_tiny_payload"] = _ktx([(4, 4, bytes(8))], w=32768, h=32768)
    files["kv_past_the_end"] = good[:60] + struct.pack("<I", 1 << 30) + good[64:]
    for k in range(400):                                              # seeded random corruption
        b = bytearray(good)
        for _ in range(int(rng.integers(1, 6))):
            b[int(rng.integers(12, len(b)))] = int(rng.integers(0, 256))
        files["fuzz%03d" % k] = bytes(b)
    paths = []
    for name, content in files.items():
        p = tmp_path / (name + ".ktx")
        p.write_bytes(content)
        paths.append(str(p))
    paths.append(str(tmp_path / "does_not_exist.ktx"))
    env = dict(os.environ, ASAN_OPTIONS="abort_on_error=0:detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    for i in range(0, len(paths), 200):
        r = subprocess.run([ktx_san] + paths[i:i + 200], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
        assert "no sanitizer report" in r.stdout


def test_entry_points_refuse_hostile_arguments_under_sanitizers():
    """tests/host_san/api_san_main.cpp: the host side of the library itself (argument validation of every entry point, the
    error convention, the half-float table builder) built with hipcc and both sanitizers on the host code (`make api-san`:
    every translation unit of the library, instrumented, linked with the test's main) and called with arguments that must be
    refused.  Works without a GPU (what passes validation then fails with "no usable HIP device").  The build takes about half a
    minute and is cached (make)."""
    exe = os.path.join(SAN, "api_san")
    subprocess.check_call(["make", "-s", "-j8", "-C", ROOT, "api-san"], stderr=subprocess.DEVNULL)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")   # (the HIP runtime keeps its own allocations)
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "0 problems, no sanitizer report" in r.stdout, (r.stdout[-3000:], r.stderr[-4000:])


def test_teardown_program_under_sanitizers_without_a_device():
    """tests/host_san/teardown_san_main.cpp instrumented (`make teardown-san`): in the container every mode must end in its "no HIP
    device" exit (code 4) -- the library loaded, initialised as far as it goes, its statics and thread_local contexts torn down, no
    sanitizer report.  With a device the uninstrumented build runs the real thing (tests/test_gpu_host_programs.py)."""
    subprocess.check_call(["make", "-s", "-j8", "-C", ROOT, "teardown-san"], stderr=subprocess.DEVNULL)
    exe = os.path.join(SAN, "teardown_san")
    lib = os.path.join(SAN, "libdetexhip_san.so")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    for mode in ("threads", "dlclose"):
        r = subprocess.run([exe, mode] + ([lib] if mode == "dlclose" else []), capture_output=True, text=True, env=env, timeout=300)
        if r.returncode == 0:                                         # a box with a device: the real run
            assert "teardown_san: ok" in r.stdout, (r.stdout[-3000:], r.stderr[-4000:])
        else:
            assert r.returncode == 4 and "no HIP device" in r.stdout and "Sanitizer" not in r.stderr, (r.returncode, r.stdout[-3000:], r.stderr[-4000:])
