"""The BPTC partition / anchor / weight constants this repository carries, in all their representations, against
the tables of the compiled reference (bptc-tables.c:23-201, exported by oracle/_ref as detex_bptc_table_*):

  detex_amd/csrc/bptc_tables.inc   bit-packed words the HIP kernels use (DETEXHIP_P2_WORDS / P2X / P3 / ANCHOR)
  oracle/bptc_partitions.inc       digit strings + anchor arrays the CPU restatement uses
  the BC7 partition-route table    derived at compile time in decode_bptc.h from those words (anchor insertion
                                   positions, window split) -- checked here through a host build of that header
  interpolation weights            closed form (64*i + (2^n-1)/2) / (2^n-1), dev_common.h bptc_weight

Without oracle/_ref (the GPU box has it prebuilt; a bare checkout does not) the two .inc files are still checked
against each other."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _words(macro):
    text = open(os.path.join(ROOT, "detex_amd", "csrc", "bptc_tables.inc")).read()
    body = re.search(r"#define %s \\\n((?:.*\\\n)*.*)\n" % macro, text).group(1)
    return [int(x, 16) for x in re.findall(r"0x([0-9A-Fa-f]+)u", body)]


def _oracle_inc():
    text = open(os.path.join(ROOT, "oracle", "bptc_partitions.inc")).read()
    strings = re.findall(r'"([012]{16})"', text)
    assert len(strings) == 128
    anchors = {name: [int(v) for v in vals.split(",")] for name, vals in re.findall(r"(orc_anchor\w+)\[64\] = \{([^}]*)\}", text)}
    return strings[:64], strings[64:], anchors


def _reference_tables():
    lib = ctypes.CDLL(ol.REF_SO)
    arr = lambda name, n, t=ctypes.c_uint8: list((t * n).in_dll(lib, name))
    return {"P2": arr("detex_bptc_table_P2", 1024), "P3": arr("detex_bptc_table_P3", 1024),
            "A2": arr("detex_bptc_table_anchor_index_second_subset", 64),
            "A3a": arr("detex_bptc_table_anchor_index_second_subset_of_three", 64),
            "A3b": arr("detex_bptc_table_anchor_index_third_subset", 64),
            "W": {2: arr("detex_bptc_table_aWeight2", 4, ctypes.c_uint16), 3: arr("detex_bptc_table_aWeight3", 8, ctypes.c_uint16),
                  4: arr("detex_bptc_table_aWeight4", 16, ctypes.c_uint16)}}


def test_device_and_oracle_tables_agree():
    p2, p2x, p3, an = _words("DETEXHIP_P2_WORDS"), _words("DETEXHIP_P2X_WORDS"), _words("DETEXHIP_P3_WORDS"), _words("DETEXHIP_ANCHOR_WORDS")
    s2, s3, anchors = _oracle_inc()
    assert [len(x) for x in (p2, p2x, p3, an)] == [64, 64, 64, 64]
    for s in range(64):
        assert [(p2[s] >> i) & 1 for i in range(16)] == [int(c) for c in s2[s]]
        assert [(p2x[s] >> (2 * i)) & 3 for i in range(16)] == [int(c) for c in s2[s]]
        assert [(p3[s] >> (2 * i)) & 3 for i in range(16)] == [int(c) for c in s3[s]]
        assert an[s] == anchors["orc_anchor2"][s] | (anchors["orc_anchor3_1"][s] << 4) | (anchors["orc_anchor3_2"][s] << 8)
        # format facts: texel 0 is always in subset 0 (it is subset 0's anchor); an anchor lies in its own subset
        assert s2[s][0] == "0" and s3[s][0] == "0"
        assert s2[s][anchors["orc_anchor2"][s]] == "1"
        assert s3[s][anchors["orc_anchor3_1"][s]] == "1" and s3[s][anchors["orc_anchor3_2"][s]] == "2"


@pytest.mark.skipif(not ol.have_ref(), reason="needs oracle/_ref")
def test_tables_equal_the_reference():
    ref = _reference_tables()
    s2, s3, anchors = _oracle_inc()
    for s in range(64):
        assert [int(c) for c in s2[s]] == ref["P2"][16 * s:16 * s + 16], s
        assert [int(c) for c in s3[s]] == ref["P3"][16 * s:16 * s + 16], s
    assert anchors["orc_anchor2"] == ref["A2"] and anchors["orc_anchor3_1"] == ref["A3a"] and anchors["orc_anchor3_2"] == ref["A3b"]
    for n, w in ref["W"].items():       # closed form used by the kernels and the oracle
        d = (1 << n) - 1
        assert w == [(64 * i + d // 2) // d for i in range(1 << n)]


def test_bc7_partition_route_table(tmp_path):
    """the compile-time-derived BC7 table (decode_bptc.h: kBc7PartTable): for every (partition table, index width) entry,
    inserting zero bits at the recorded positions turns the irregular index stream (anchors one bit short) into a
    regular one -- checked against a direct computation from the anchor arrays"""
    src = tmp_path / "route.cpp"
    src.write_text(r'''
#include "hip_host_shim.h"
#include "dev_common.h"
#include "decode_bptc.h"
using namespace detexhip;
extern "C" void dump(uint32_t *out) { for (int i = 0; i < kBc7PartEntries; i++) { out[2 * i] = kBc7PartTable.e[i].pword; out[2 * i + 1] = kBc7PartTable.e[i].route; } }
extern "C" int entries(void) { return kBc7PartEntries; }
''')
    so = tmp_path / "route.so"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "tests", "host_emul"),
                           "-I" + os.path.join(ROOT, "detex_amd", "csrc"), "-Wno-unknown-pragmas", "-o", str(so), str(src)])
    lib = ctypes.CDLL(str(so))
    n = lib.entries()
    buf = (ctypes.c_uint32 * (2 * n))()
    lib.dump(buf)
    table = np.array(buf, np.uint32).reshape(n, 2)
    s2, s3, anchors = _oracle_inc()
    sections = [(0, 64, 2, 2), (64, 64, 2, 3), (128, 64, 3, 2), (192, 16, 3, 3)]
    for base, count, subsets, ib in sections:
        for p in range(count):
            pword, route = int(table[base + p][0]), int(table[base + p][1])
            want = s2[p] if subsets == 2 else s3[p]
            assert [(pword >> (2 * i)) & 3 for i in range(16)] == [int(c) for c in want]
            an = [0, anchors["orc_anchor2"][p]] if subsets == 2 else [0, anchors["orc_anchor3_1"][p], anchors["orc_anchor3_2"][p]]
            lo = sorted((a & 7) * ib + ib - 1 for a in an[1:] if a < 8) + [31, 31]
            hi = sorted((a & 7) * ib + ib - 1 for a in an[1:] if a >= 8) + [31, 31]
            assert [route & 31, (route >> 5) & 31] == lo[:2] and [(route >> 10) & 31, (route >> 15) & 31] == hi[:2], (base, p)
            assert (route >> 20) == 8 * ib - sum(1 for a in an if a < 8)
    for k, ib in ((208, 2), (209, 3), (210, 4)):
        assert int(table[k][0]) == 0 and (int(table[k][1]) >> 20) == 8 * ib - 1
