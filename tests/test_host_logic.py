"""CPU tests of the host-side logic and of the DEVICE decode logic under emulation.

* test_intmath: the multiply-shift divisions / weights of detex_amd/csrc/dev_common.h are
  exhaustively equal to C integer division on the domains of the reference's LUTs.
* test_device_logic_under_emulation: detex_amd/csrc/decode_*.h compiled with g++ against
  tests/host_emul/hip_host_shim.h (amdgcn builtins, one emulated lane, gfx950_prims.h in plain C++) and compared with the
  oracle on every mode-forced class x the mask/flag matrix + random blocks.  This catches logic
  errors in this GPU-less container; it is NOT a parity claim for the hardware path (that is
  tests/test_gpu_parity.py, which has already caught a code-generation problem emulation cannot
  see).  Nothing under tests/host_emul is part of libdetexhip.so.
* test_sharding_*: shard arithmetic, and the N>1 path over gloo with world_size 2.
"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as ol
import streams
from detex_amd import formats as F, sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMUL = os.path.join(ROOT, "tests", "host_emul")
CSRC = os.path.join(ROOT, "detex_amd", "csrc")
FMT_IDS = [f.name for f in F.FORMATS]


def _compile(src, out, extra=()):
    deps = [src] + [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(EMUL, "hip_host_shim.h")]
    if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
                           "-Wno-unknown-pragmas", "-I" + EMUL, "-I" + CSRC, "-o", out, src] + list(extra))
    return out


def test_intmath(tmp_path):
    src = tmp_path / "intmath.cpp"
    src.write_text(r'''
#include "dev_common.h"
using namespace detexhip;
extern "C" int check(void) {
	for (uint32_t x = 0; x < 4096; x++) {
		if (div3_u(x) != x / 3) return 3;
		if (div5_u(x) != x / 5) return 5;
		if (div7_u(x) != x / 7) return 7;
	}
	for (int v = -2047; v <= 2047; v++) {
		if (div7_s(v) != v / 7) return 17;
		if (div5_s(v) != v / 5) return 15;
	}
	for (uint32_t bits = 2; bits <= 4; bits++)
		for (uint32_t i = 0; i < (1u << bits); i++) {
			const uint32_t d = (1u << bits) - 1;
			if (bptc_weight(i, bits) != (64 * i + d / 2) / d) return 100 + bits;
		}
	for (uint32_t x = 0; x < 65536; x++)		// 16 -> 8 bit component of the conversion epilogues (convert.c:258-267)
		if (component16_to_8(x) != (x + 127) * 255 / 65535) return 150;
	for (int v = -127; v <= 127; v++)
		if (rgtc_signed_to_16(v) != (uint32_t)(uint16_t)(int16_t)((v + 127) * 65535 / 254 - 32768)) return 200;
	// the biased forms used by the signed RGTC decoder (decode_s3tc_rgtc.h: rgtc_channel_s16)
	for (uint32_t n = 0; n <= 254; n++) {
		if (((n * 387u + 6u) >> 15) != 3 * n / 254) return 201;
		if (((n * 258u + ((n * 387u + 6u) >> 15)) ^ 0x8000u) != rgtc_signed_to_16((int)n - 127)) return 202;
	}
	for (int x = -127 * 7; x <= 127 * 7; x++)
		if ((int)div7_u((uint32_t)(x + 896 + ((x >> 31) & 6))) - 128 != x / 7) return 203;
	for (int x = -127 * 5; x <= 127 * 5; x++)
		if ((int)div5_u((uint32_t)(x + 640 + ((x >> 31) & 4))) - 128 != x / 5) return 204;
	// BPTC index -> weight as one multiply-add: byte 2 of (64*i + d/2) * ceil(65536/d) (decode_bptc.h: weight_mad)
	{
		const uint32_t mul[5] = { 0, 0, 1398144u, 599232u, 279680u }, add[5] = { 0, 0, 21846u, 28089u, 30590u };
		for (uint32_t bits = 2; bits <= 4; bits++)
			for (uint32_t i = 0; i < (1u << bits); i++) {
				const uint32_t t = i * mul[bits] + add[bits];
				if (t >> 24 || ((t >> 16) & 0xFFu) != bptc_weight(i, bits)) return 210 + bits;
			}
	}
	// ... and in 16 bits, two indices per register (decode_bptc.h, both index streams in lockstep): (i * m + 128) >> 8, no 16-bit overflow
	for (uint32_t bits = 2; bits <= 4; bits++)
		for (uint32_t i = 0; i < (1u << bits); i++) {
			const uint32_t t = i * bptc_weight16_mul(bits) + 128u;
			if (t >> 16 || (t >> 8) != bptc_weight(i, bits)) return 220 + bits;
		}
	return 0;
}
''')
    lib = ctypes.CDLL(_compile(str(src), str(tmp_path / "intmath.so")))
    assert lib.check() == 0


@pytest.fixture(scope="module")
def emul():
    lib = ctypes.CDLL(_compile(os.path.join(EMUL, "emul_decoders.cpp"), os.path.join(EMUL, "libemul.so")))
    u8p = ctypes.POINTER(ctypes.c_uint8)
    lib.emul_decode_blocks.argtypes = [ctypes.c_int, u8p, ctypes.c_long, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, u8p, u8p]

    lib.emul_set_other_lanes_vote.argtypes = [ctypes.c_ulonglong]

    def run(fmt, blocks, mask=0xFFFFFFFF, flags=0, checked=1, mixed_wave=False):
        """mixed_wave: every wave-uniform vote answers yes -- each block decoded as inside a wave that holds every kind of block"""
        lib.emul_set_other_lanes_vote(0xFFFFFFFFFFFFFFFE if mixed_wave else 0)
        blocks = np.ascontiguousarray(blocks, np.uint8).reshape(-1)
        n = blocks.size // fmt.block_bytes
        out = np.zeros((n, 16 * fmt.pixel_bytes), np.uint8)
        ok = np.zeros(n, np.uint8)
        assert lib.emul_decode_blocks(fmt.index, ol._ptr(blocks), n, mask, flags, checked, ol._ptr(out), ol._ptr(ok)) == 0
        return ok.astype(bool), out
    return run


@pytest.mark.parametrize("fmt", F.FORMATS, ids=FMT_IDS)
def test_device_logic_under_emulation(fmt, emul, oracle, forced_vectors):
    blocks = np.concatenate([forced_vectors[fmt.name + "/in"],
                             ol.stream_u(fmt, 1 << 14, seed=0x51DE + fmt.index).reshape(-1, fmt.block_bytes)])
    for mask, flags in streams.MASK_FLAG_MATRIX:
        ok_e, out_e = emul(fmt, blocks, mask, flags, checked=1)
        ok_o, out_o = oracle.blocks(fmt, blocks, mask, flags)
        assert np.array_equal(ok_e, ok_o), (fmt.name, hex(mask), flags)
        assert np.array_equal(out_e, out_o), (fmt.name, hex(mask), flags)
    ok_o, out_o = oracle.blocks(fmt, blocks)
    for mixed in (False, True):                         # alone in its wave / inside a wave that holds every kind of block
        ok_e, out_e = emul(fmt, blocks, checked=0, mixed_wave=mixed)          # the texture-driver instantiation (mask ALL, flags 0 folded away)
        assert np.array_equal(ok_e, ok_o) and np.array_equal(out_e, out_o), (fmt.name, mixed)


@pytest.mark.parametrize("alt,name", [(109, "BPTC_FLOAT"), (110, "BPTC_SIGNED_FLOAT"), (111, "BPTC")])
def test_alternative_decoders_under_emulation(alt, name, emul, oracle, forced_vectors):
    """The A/B decoder implementations behind detexhipSetKernelVariant(3/4) decode identically."""
    import types
    fmt = F.BY_NAME[name]
    blocks = np.concatenate([forced_vectors[name + "/in"], ol.stream_u(fmt, 1 << 14, seed=0xA17 + alt).reshape(-1, fmt.block_bytes)])
    shim = types.SimpleNamespace(index=alt, block_bytes=fmt.block_bytes, pixel_bytes=fmt.pixel_bytes)
    for checked in (1, 0):
        ok_e, out_e = emul(shim, blocks, checked=checked)
        ok_o, out_o = oracle.blocks(fmt, blocks)
        assert np.array_equal(ok_e, ok_o) and np.array_equal(out_e, out_o)


@pytest.mark.parametrize("index,name,spec_flag,quirk", [(11, "BPTC", 1 << 30, 1), (111, "BPTC", 1 << 30, 1), (9, "BPTC_FLOAT", 1 << 31, 2), (10, "BPTC_SIGNED_FLOAT", 1 << 31, 2),
                                                         (109, "BPTC_FLOAT", 1 << 31, 2), (110, "BPTC_SIGNED_FLOAT", 1 << 31, 2)])
def test_spec_switches_under_emulation(index, name, spec_flag, quirk, emul, oracle, forced_vectors):
    """the decoders' spec-conformance flags (bptc_common.h: kFlagSpec..., set by detexhipSetQuirks) against the checker with the
    corresponding quirk off (tests/test_quirks.py pins that switch)"""
    import types
    fmt = F.BY_NAME[name]
    blocks = np.concatenate([forced_vectors[name + "/in"], ol.stream_u(fmt, 1 << 14, seed=0x5EC + index).reshape(-1, fmt.block_bytes)])
    shim = types.SimpleNamespace(index=index, block_bytes=fmt.block_bytes, pixel_bytes=fmt.pixel_bytes)
    oracle.lib.orc_set_quirks.argtypes = [ctypes.c_uint]
    try:
        oracle.lib.orc_set_quirks(3 & ~quirk)
        ok_o, out_o = oracle.blocks(fmt, blocks)
        for checked in (1, 0):
            ok_e, out_e = emul(shim, blocks, flags=spec_flag, checked=checked)
            assert np.array_equal(ok_e, ok_o) and np.array_equal(out_e, out_o)
        oracle.lib.orc_set_quirks(3)
        ok_q, out_q = oracle.blocks(fmt, blocks)
        assert (out_q != out_o).any(), "the forced vectors contain blocks the quirk changes"
    finally:
        oracle.lib.orc_set_quirks(3)


def test_signed_bc6h_extreme_magnitudes_under_emulation(emul, oracle):
    """mode-13 blocks whose interpolated value reaches -32768 (sign-magnitude half 0xFC00): the packed 16-bit
    finish of the signed BC6H kernel must treat |v| = 0x8000 as unsigned (tests/golden/bc6h_signed_extreme_blocks.npy:
    64 blocks found by random search, expected values from the oracle)"""
    fmt = F.BY_NAME["BPTC_SIGNED_FLOAT"]
    blocks = np.load(os.path.join(ROOT, "tests", "golden", "bc6h_signed_extreme_blocks.npy"))
    ok_o, out_o = oracle.blocks(fmt, blocks)
    assert (out_o.view(np.uint16) == 0xFC00).any(axis=1).all()
    ok_e, out_e = emul(fmt, blocks)
    assert np.array_equal(ok_e, ok_o) and np.array_equal(out_e, out_o)


# ---- sharding -----------------------------------------------------------------------------------------
def test_shard_arithmetic_tiles_the_texture_exactly():
    for fmt in (F.BY_NAME["BC1"], F.BY_NAME["BPTC_FLOAT"], F.BY_NAME["RGTC1"]):
        for (w, h) in ((8192, 8192), (32768, 32768), (100, 36), (64, 4), (7, 13)):
            wb, hb = (w + 3) // 4, (h + 3) // 4
            for world in (1, 2, 3, 4, 8):
                shards = [sharding.shard_of(r, world, fmt, w, h) for r in range(world)]
                assert shards[0].in_offset == 0 and shards[0].out_offset == 0
                for a, b in zip(shards, shards[1:]):
                    assert a.in_offset + a.in_bytes == b.in_offset and a.row1 == b.row0
                    assert a.out_offset + a.out_bytes == b.out_offset
                assert shards[-1].in_offset + shards[-1].in_bytes == wb * hb * fmt.block_bytes
                assert shards[-1].out_offset + shards[-1].out_bytes == w * h * fmt.pixel_bytes
                assert max(s.row1 - s.row0 for s in shards) - min(s.row1 - s.row0 for s in shards) <= 1
    s = sharding.shard_of(3, 8, F.BY_NAME["BPTC_FLOAT"], 32768, 32768)
    assert s.in_bytes == 128 << 20 and s.out_bytes == 1 << 30          # 128 MiB in, 1 GiB out per GPU


def _gloo_worker(rank, world, port, q):
    try:
        import torch
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        orc = ol.Oracle()
        results = []
        for name, (w, h) in (("BC1", (256, 200)), ("BPTC", (64, 36)), ("EAC_R11", (100, 52))):
            fmt = F.BY_NAME[name]
            wb, hb = (w + 3) // 4, (h + 3) // 4
            data = ol.stream_u(fmt, wb * hb, seed=4242 + fmt.index)

            def decode(fmt_, band, width, rows):          # stands in for the device-tier call on CPU
                ok, px = orc.linear(fmt_, np.ascontiguousarray(band), width, rows)
                return ok, torch.from_numpy(px)
            shard, ok, local = sharding.decode_shard(decode, fmt, data, w, h, rank, world)
            ok_all, image = sharding.gather_image(dist, torch, fmt, w, h, shard, local, ok)
            ok_ref, want = orc.linear(fmt, data, w, h)
            results.append((name, ok_all == ok_ref, bool(np.array_equal(image.numpy(), want))))
            # the same image gathered to one rank only (the last one, to exercise root != 0): grouped sends / receives
            root = world - 1
            ok_root, image_root = sharding.gather_image_to_root(dist, torch, fmt, w, h, shard, local, ok, root=root)
            results.append((name + " to root", ok_root == ok_ref,
                            bool(np.array_equal(image_root.numpy(), want)) if rank == root else image_root is None))
        if world >= 4:
            # a sub-group whose group ranks differ from the global ones (global 1 and 3 are its ranks 0 and 1): the image is sharded over
            # the GROUP and gathered to the group's rank 1 (= global 3); point-to-point peers must be translated to global ranks
            members = [1, 3]
            sub = dist.new_group(members, backend="gloo")          # (every rank of the default group takes part in the creation)
            if rank in members:
                fmt, (w, h) = F.BY_NAME["BC3"], (64, 40)
                data = ol.stream_u(fmt, 16 * 10, seed=99)
                grank = members.index(rank)
                shard, ok, local = sharding.decode_shard(lambda f, band, width, rows: (lambda r: (r[0], torch.from_numpy(r[1])))(orc.linear(f, np.ascontiguousarray(band), width, rows)),
                                                         fmt, data, w, h, grank, len(members))
                ok_root, image_root = sharding.gather_image_to_root(dist, torch, fmt, w, h, shard, local, ok, root=1, group=sub)
                ok_ref, want = orc.linear(fmt, data, w, h)
                results.append(("sub-group to root", ok_root == ok_ref, bool(np.array_equal(image_root.numpy(), want)) if grank == 1 else image_root is None))
                ok_all, image = sharding.gather_image(dist, torch, fmt, w, h, shard, local, ok, group=sub)
                results.append(("sub-group to all", ok_all == ok_ref, bool(np.array_equal(image.numpy(), want))))
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, results))
    except Exception as e:   # noqa
        q.put((rank, "ERROR %r" % (e,)))


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_decode_and_gather_over_gloo(world):
    """the N>1 path (row-band shards + optional whole-image gather) with two / four CPU processes; with four ranks the 9-,
    13- and 50-block-row images give bands of unequal height (the padded-chunk branch of gather_image)"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + 7 * world
    procs = [ctx.Process(target=_gloo_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    got = [q.get(timeout=180) for _ in procs]
    for p in procs: p.join(timeout=60)
    for rank, results in got:
        assert not isinstance(results, str), results
        for name, ok_match, image_match in results:
            assert ok_match and image_match, (rank, name)
