"""C-ABI tests that need no GPU: libdetexhip.so loads, exports every symbol include/*.h declares,
its data tables hold the right values, calls fail LOUDLY without a device (no CPU fallback), and a
client compiled against the reference's own detex.h links against it (build container only)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

import oracle_lib as ol
from detex_amd import binding, formats as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = set()
    for hdr in ("detex.h", "detexhip.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        names |= set(re.findall(r"\b(detexhip[A-Z]\w+)\s*\(", text))
        names |= set(re.findall(r"DETEX_API\s+[\w\s\*]*?\b(detex[A-Z]\w+)\s*\(", text))
        names |= {"detexDecompressBlock" + n for n in re.findall(r"DETEXHIP_DECLARE_BLOCK_FN\((\w+)\)\s*/\*", text)}
        names |= set(re.findall(r"extern const uint8_t (detex_\w+)\[", text))
    return names


def test_library_exports_every_declared_symbol():
    lib = binding.load()
    declared = _declared_symbols()
    assert len(declared) == (19 + 3 + 2 + 2) + (9 + 4 + 10 + 2 + 4 + 2) + 4, sorted(declared)   # detex.h, detexhip.h (+ multi-device incl. the host-output entry, release, host aliases, half table, accumulating histogram, quirk switch, the two resident-service calls; round 5: the batched host-pointer block entry, the ABI check, the pixel-buffer allocator pair; round 6: the fail-after-launch test hook, the read-ahead switch), data tables
    out = subprocess.check_output(["nm", "-D", "--defined-only", binding.LIB_PATH], text=True)
    exported = {line.split()[-1] for line in out.splitlines() if line.strip()}
    assert declared <= exported, sorted(declared - exported)
    # nothing of the checker leaks into the product
    assert not any(s.startswith("orc_") for s in exported)
    assert lib.detexhipVersion().decode().startswith("libdetexhip")
    # the extension ABI's version check (struct layouts of detexhip.h): the header's number is accepted, any other refused with a message
    abi = int(re.search(r"#define DETEXHIP_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "detexhip.h")).read()).group(1))
    assert lib.detexhipCheckAbi(abi) == 0 and lib.detexhipCheckAbi(abi - 1) != 0 and "extension ABI" in binding.last_error()
    assert ("extension ABI %d" % abi) in lib.detexhipVersion().decode()


def test_product_does_not_link_the_oracle():
    deps = subprocess.check_output(["readelf", "-d", binding.LIB_PATH], text=True)
    assert "oracle" not in deps and "detex_ref" not in deps
    for root, _, files in os.walk(os.path.join(ROOT, "detex_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".inc")):
                text = open(os.path.join(root, f)).read()
                assert "oracle/" not in text.replace("oracle/_ref", "").replace("oracle/bptc_partitions.inc", "") or f in ("binding.py",), f
                assert "import oracle_lib" not in text and "libdetex_oracle" not in text, f


def test_data_tables_have_reference_semantics():
    lib = binding.load()
    t = lambda name, n: np.array((ctypes.c_uint8 * n).in_dll(lib, name))
    assert np.array_equal(t("detex_division_by_3_table", 768), np.arange(768) // 3)
    assert np.array_equal(t("detex_division_by_5_table", 1280), np.arange(1280) // 5)
    assert np.array_equal(t("detex_division_by_7_table", 1792), np.arange(1792) // 7)
    assert np.array_equal(t("detex_clamp0to255_table", 767), np.clip(np.arange(767) - 255, 0, 255))
    if ol.have_ref():   # byte-identical to the reference's LUTs
        ref = ctypes.CDLL(ol.REF_SO)
        for name, n in (("detex_division_by_3_table", 768), ("detex_division_by_5_table", 1280),
                        ("detex_division_by_7_table", 1792), ("detex_clamp0to255_table", 767)):
            assert np.array_equal(t(name, n), np.array((ctypes.c_uint8 * n).in_dll(ref, name))), name


def _have_gpu():
    return binding.load().detexhipGetDeviceCount() > 0


@pytest.mark.skipif(_have_gpu(), reason="checks the no-device behaviour")
def test_calls_fail_loudly_without_a_device():
    api = ol.DetexAPI(binding.LIB_PATH)
    fmt = F.BY_NAME["BC1"]
    ok, out = api.linear(fmt, np.zeros(16 * 16 * 8, np.uint8), 64, 64)
    assert not ok and "no CPU decode path" in api.error()
    ok, out = api.block(fmt, np.zeros(8, np.uint8))
    assert not ok
    assert binding.load().detexhipGetDeviceCount() == 0
    lib = binding.load()
    lib.detexhipAllocPixelBuffer.restype = ctypes.c_void_p
    lib.detexhipAllocPixelBuffer.argtypes = [ctypes.c_size_t]
    assert lib.detexhipAllocPixelBuffer(4096) is None and "detexhipAllocPixelBuffer" in binding.last_error()      # no pinned memory without a device


def test_argument_validation_needs_no_device():
    api = ol.DetexAPI(binding.LIB_PATH)
    out = np.zeros(64, np.uint8)
    blk = np.zeros(16, np.uint8)
    # ASTC_4X4 (index 20) and "uncompressed" (index 0) are read past / NULL-called in the reference (A-11)
    for tf in (0x14800334, 0x00000334, 0xFF000000):
        assert not api.lib.detexDecompressBlock(ol._ptr(blk), tf, 0xFFFFFFFF, 0, ol._ptr(out), 0x334)
        assert "not a block-compressed format" in api.error()
    assert not api.lib.detexDecompressBlock(ol._ptr(blk), F.BY_NAME["BC1"].texture_format, 0xFFFFFFFF, 0, ol._ptr(out), 0x228)   # BGR8: unreachable in the reference as well
    assert "outside the block-decode path" in api.error()
    lib = binding.load()
    assert lib.detexhipDecompressTextureLinearDevice(F.BY_NAME["BC1"].texture_format, None, 8, 8, 2, 2, None, 4, 0x334, None, None) != 0
    assert "bad geometry" in binding.last_error()
    # error string is replaced, not appended (misc.c:77-89)
    lib.detexSetErrorMessage(b"custom %d", 7)
    assert binding.last_error() == "custom 7"


def test_header_constants_match_the_format_table():
    text = open(os.path.join(ROOT, "include", "detex.h")).read()
    for f in F.FORMATS:
        m = re.search(r"DETEX_TEXTURE_FORMAT_%s = (0x[0-9A-Fa-f]+)," % f.name, text)
        assert m and int(m.group(1), 16) == f.texture_format, f.name


@pytest.mark.skipif(not os.path.exists("/root/reference/detex.h"), reason="needs the reference header")
def test_client_built_against_reference_header_links(tmp_path):
    """drop-in proof: an unmodified client TU compiled against /root/reference/detex.h (which inlines
    helpers that reference the data tables) links against libdetexhip.so, and our compat header
    agrees with the reference's on every constant and on the detexTexture layout."""
    src = tmp_path / "client.c"
    src.write_text(r'''
#include <stdio.h>
#include <stddef.h>
#include "detex.h"
int main(void) {
	uint8_t block[16] = {0}, out[256];
	detexTexture t = { DETEX_TEXTURE_FORMAT_BC1, block, 4, 4, 1, 1 };
	bool r = detexDecompressTextureLinear(&t, out, DETEX_PIXEL_FORMAT_RGBA8);
	r |= detexDecompressBlockBPTC(block, DETEX_MODE_MASK_ALL, 0, out);
	printf("%d %d %u %u %zu %zu\n", (int)detexClamp0To255(300), (int)detexDivide0To767By3(100),
		detexDivide0To1791By7(700), detexDivide0To1279By5(55), sizeof(detexTexture), offsetof(detexTexture, data));
	const char *m = detexGetErrorMessage();
	printf("%s\n", m ? m : "(null)");
	return r ? 0 : 0;
}
''')
    exe = tmp_path / "client"
    libdir = os.path.dirname(binding.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-D_POSIX_C_SOURCE=200809L", "-I/root/reference", str(src), "-o", str(exe),
                           "-L" + libdir, "-ldetexhip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.check_output([str(exe)], text=True).splitlines()
    assert out[0] == "255 33 100 11 32 8"
    # constants of our header == constants of the reference header
    probe = tmp_path / "probe.c"
    names = re.findall(r"\b(DETEX_(?:PIXEL_FORMAT|TEXTURE_FORMAT|MODE_MASK|DECOMPRESS_FLAG)_\w+) =", open(os.path.join(ROOT, "include", "detex.h")).read())
    body = "\n".join('printf("%s %%u\\n", (unsigned)%s);' % (n, n) for n in names)
    probe.write_text('#include <stdio.h>\n#include "detex.h"\nint main(void){%s return 0;}' % body)
    vals = {}
    for inc in ("/root/reference", os.path.join(ROOT, "include")):
        e = tmp_path / ("probe_" + str(abs(hash(inc))))
        subprocess.check_call(["gcc", "-I" + inc, str(probe), "-o", str(e)])
        vals[inc] = subprocess.check_output([str(e)], text=True)
    assert vals["/root/reference"] == vals[os.path.join(ROOT, "include")]


def test_half_float_table_matches_the_oracle_chain(oracle):
    """all 65536 half patterns: the host function the device table is built from == the restated FLOAT_RGBX16 ->
    RGBX16 -> RGBX8 chain (itself pinned to the compiled reference in tests/test_oracle_pin.py)"""
    lib = binding.load()
    lib.detexhipHalfFloatToUNorm8.restype = ctypes.c_uint8
    lib.detexhipHalfFloatToUNorm8.argtypes = [ctypes.c_uint16]
    got = np.array([lib.detexhipHalfFloatToUNorm8(h) for h in range(65536)], np.uint8)
    fmt = F.BY_NAME["BPTC_FLOAT"]
    src = np.zeros((65536, 4), np.uint16)
    src[:, 0] = np.arange(65536)
    want = oracle.convert(fmt, src.view(np.uint8).reshape(-1), F.PIXEL_FORMAT_RGBX8).reshape(-1, 4)[:, 0]
    assert np.array_equal(got, want), np.flatnonzero(got != want)[:10]
