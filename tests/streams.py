"""Deterministic block-stream builders for the parity tests: mode-forced classes for every
format, clipped texture sizes, and the (mode_mask, flags) matrix of
the per-block API.  Pure numpy; the classes are *constructed* (bit patterns forced), and where
a class cannot be forced by bit-twiddling (ETC2 T/H/planar, ETC1 overflow) it is rejection-
sampled with a classifier written here from the format definition (not the oracle)."""
import numpy as np

from detex_amd import formats as F

N_PER_CLASS = 256

CLIP_SIZES = [(1, 1), (2, 3), (5, 9), (10, 6), (7, 13), (16, 4), (33, 17), (64, 64), (100, 36)]
CLIP_SIZES_CONVERTED = [(5, 9), (16, 4), (33, 17), (100, 36)]     # also decoded into the epilogue targets

# (mode_mask, flags) combinations exercised through the per-block API
MASK_FLAG_MATRIX = [
    (0xFFFFFFFF, 0), (0xFFFFFFFF, F.FLAG_ENCODE), (0xFFFFFFFF, F.FLAG_OPAQUE_ONLY),
    (0xFFFFFFFF, F.FLAG_NON_OPAQUE_ONLY), (0x1, 0), (0x2, 0), (0x4, 0), (0x8, 0), (0x10, 0),
    (0x1E, F.FLAG_OPAQUE_ONLY), (0x55, 0), (0xAA, F.FLAG_NON_OPAQUE_ONLY), (0x1555, 0), (0x2AAA, 0), (0, 0),
]


def _rng(tag):
    return np.random.default_rng(abs(hash_str(tag)) % (1 << 63))


def hash_str(s):
    h = 0xcbf29ce484222325
    for ch in s.encode():
        h = ((h ^ ch) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h


def random_blocks(tag, n, block_bytes):
    return _rng(tag).integers(0, 256, size=(n, block_bytes), dtype=np.uint8)


def _etc_class(b):
    """0 individual, 1 differential, 2 T, 3 H, 4 planar for ETC2-style colour words b[n,8];
    5-bit base + 3-bit two's-complement delta out of 0..31 selects T (red), H (green), planar (blue)."""
    def ovf(x):
        base = (x >> 3).astype(np.int32)
        d = (x & 7).astype(np.int32)
        d = np.where(d >= 4, d - 8, d)
        s = base + d
        return (s < 0) | (s > 31)
    diff = (b[:, 3] & 2) != 0
    r, g, bl = ovf(b[:, 0]), ovf(b[:, 1]), ovf(b[:, 2])
    cls = np.where(r, 2, np.where(g, 3, np.where(bl, 4, 1)))
    return np.where(diff, cls, 0), cls


def _sample(tag, block_bytes, pred, n=N_PER_CLASS, colour_off=0):
    """rejection-sample n blocks whose 8-byte colour word (at colour_off) satisfies pred."""
    out = []
    have = 0
    rnd = 0
    while have < n:
        b = random_blocks("%s/%d" % (tag, rnd), 64 * n, block_bytes)
        keep = b[pred(b[:, colour_off:colour_off + 8])]
        out.append(keep)
        have += len(keep)
        rnd += 1
        assert rnd < 200, tag
    return np.concatenate(out)[:n]


def _bc_order(b, off, want_greater):
    """force c0 > c1 (or c0 <= c1) of the little-endian RGB565 pair at byte offset off."""
    b = b.copy()
    c = b[:, off:off + 4].copy().view("<u2")
    c0, c1 = c[:, 0].copy(), c[:, 1].copy()
    gt = c0 > c1
    swap = gt != want_greater
    if want_greater:
        swap |= (c0 == c1)
    c[swap, 0], c[swap, 1] = c1[swap], c0[swap]
    if want_greater:                       # equal pairs cannot be made "greater" by swapping
        eq = c[:, 0] == c[:, 1]
        c[eq, 0] = np.maximum(c[eq, 0], 1)
        c[eq, 1] = c[eq, 0] - 1
    b[:, off:off + 4] = c.view(np.uint8)
    return b


BC6H_MODE_CODES = [0x00, 0x01, 0x02, 0x06, 0x0A, 0x0E, 0x12, 0x16, 0x1A, 0x1E, 0x03, 0x07, 0x0B, 0x0F]
BC6H_RESERVED = [0x13, 0x17, 0x1B, 0x1F]


def forced_classes(fmt):
    """-> list of (label, blocks[n, block_bytes]) covering every mode / invalid class of fmt."""
    n, bs, name = N_PER_CLASS, fmt.block_bytes, fmt.name
    R = lambda label: random_blocks("%s/%s" % (name, label), n, bs)
    out = [("random", R("random"))]
    if name in ("BC1", "BC1A"):
        out += [("c0_gt_c1", _bc_order(R("gt"), 0, True)), ("c0_le_c1", _bc_order(R("le"), 0, False))]
        eq = R("eq"); eq[:, 2:4] = eq[:, 0:2]; out.append(("c0_eq_c1", eq))
    elif name in ("BC2", "BC3"):
        out += [("c0_gt_c1", _bc_order(R("gt"), 8, True)), ("c0_le_c1", _bc_order(R("le"), 8, False))]
        if name == "BC3":
            a = R("a_gt"); a[:, 0] = np.maximum(a[:, 0], 1); a[:, 1] = a[:, 0] - 1 - (a[:, 1] % a[:, 0]); out.append(("a0_gt_a1", a))
            a = R("a_le"); a[:, 1] = np.maximum(a[:, 0], a[:, 1]); out.append(("a0_le_a1", a))
            a = R("a_ext"); a[:, 0] = np.where(np.arange(n) % 2, 255, 0); a[:, 1] = np.where(np.arange(n) % 4 < 2, 255, 0); out.append(("a_extremes", a))
    elif name in ("RGTC1", "RGTC2", "SIGNED_RGTC1", "SIGNED_RGTC2"):
        for ch in range(bs // 8):
            o = 8 * ch
            a = R("gt%d" % ch); a[:, o] = np.maximum(a[:, o], 1).astype(np.uint8)
            if name.startswith("SIGNED"):
                s0 = a[:, o].view(np.int8).astype(np.int32); s1 = a[:, o + 1].view(np.int8).astype(np.int32)
                lo, hi = np.minimum(s0, s1), np.maximum(s0, s1)
                a[:, o] = hi.astype(np.int8).view(np.uint8); a[:, o + 1] = lo.astype(np.int8).view(np.uint8)
                out.append(("e0_ge_e1_ch%d" % ch, a))
                b = a.copy(); b[:, o], b[:, o + 1] = a[:, o + 1], a[:, o]; out.append(("e0_le_e1_ch%d" % ch, b))
                inv = R("inv%d" % ch); inv[:, o] = 0x81; inv[:, o + 1] = 0x80; out.append(("invalid_pair_ch%d" % ch, inv))
                m = R("m128_%d" % ch); m[:, o] = 0x80; out.append(("e0_minus128_ch%d" % ch, m))
                m = R("m128b_%d" % ch); m[:, o + 1] = 0x80; out.append(("e1_minus128_ch%d" % ch, m))
                m = R("both128_%d" % ch); m[:, o] = 0x80; m[:, o + 1] = 0x80; out.append(("both_minus128_ch%d" % ch, m))
            else:
                lo, hi = np.minimum(a[:, o], a[:, o + 1]), np.maximum(a[:, o], a[:, o + 1])
                a[:, o], a[:, o + 1] = hi, lo; out.append(("e0_ge_e1_ch%d" % ch, a))
                b = a.copy(); b[:, o], b[:, o + 1] = lo, hi; out.append(("e0_le_e1_ch%d" % ch, b))
    elif name in ("ETC1", "ETC2", "ETC2_PUNCHTHROUGH", "ETC2_EAC"):
        co = 8 if name == "ETC2_EAC" else 0
        for flip in (0, 1):
            def with_flip(p, flip=flip):
                return lambda c: p(c) & ((c[:, 3] & 1) == flip)
            if name != "ETC2_PUNCHTHROUGH":
                out.append(("individual_flip%d" % flip, _sample("%s/ind%d" % (name, flip), bs, with_flip(lambda c: _etc_class(c)[0] == 0), colour_off=co)))
            out.append(("differential_flip%d" % flip, _sample("%s/diff%d" % (name, flip), bs, with_flip(lambda c: (_etc_class(c)[1] == 1) & ((c[:, 3] & 2) != 0)), colour_off=co)))
        labels = {2: "T", 3: "H", 4: "planar"}
        for k, lab in labels.items():
            if name == "ETC1":
                out.append(("overflow_%s" % lab, _sample("%s/ovf%d" % (name, k), bs, lambda c, k=k: _etc_class(c)[0] == k, colour_off=co)))
            else:
                out.append((lab, _sample("%s/%s" % (name, lab), bs, lambda c, k=k: (_etc_class(c)[1] == k) & ((c[:, 3] & 2) != 0), colour_off=co)))
        if name == "ETC2_PUNCHTHROUGH":
            for k, lab in {1: "differential", 2: "T", 3: "H", 4: "planar"}.items():
                out.append(("nonopaque_%s" % lab, _sample("%s/no%d" % (name, k), bs, lambda c, k=k: (_etc_class(c)[1] == k) & ((c[:, 3] & 2) == 0))))
        if name == "ETC2_EAC":
            m = R("mult0"); m[:, 1] &= 0x0F; out.append(("alpha_multiplier0", m))
            m = R("ext"); m[:, 0] = np.where(np.arange(n) % 2, 255, 0); m[:, 1] |= 0xF0; out.append(("alpha_extremes", m))
    elif name.startswith("EAC_"):
        for ch in range(bs // 8):
            o = 8 * ch
            m = R("mult0_%d" % ch); m[:, o + 1] &= 0x0F; out.append(("multiplier0_ch%d" % ch, m))
            m = R("mult15_%d" % ch); m[:, o + 1] |= 0xF0; m[:, o] = np.where(np.arange(n) % 2, 0xFF if "SIGNED" not in name else 0x7F, 0x00 if "SIGNED" not in name else 0x81); out.append(("saturating_ch%d" % ch, m))
            if "SIGNED" in name:
                m = R("base128_%d" % ch); m[:, o] = 0x80; out.append(("base_minus128_ch%d" % ch, m))
    elif name == "BPTC":
        for mode in range(8):
            b = R("mode%d" % mode)
            b[:, 0] = (b[:, 0] & np.uint8((0xFF << (mode + 1)) & 0xFF)) | np.uint8(1 << mode)
            out.append(("mode%d" % mode, b))
        b = R("reserved"); b[:, 0] = 0; out.append(("reserved", b))
        # all 64 partitions x both index-selection / all rotations appear in 256 random blocks; add
        # endpoint extremes for the interpolation rounding
        for mode in range(8):
            b = np.where(np.arange(n)[:, None] % 2, 0xFF, 0x00).astype(np.uint8) * np.ones((n, bs), np.uint8)
            b[:, 12:] = R("ext%d" % mode)[:, 12:]
            b[:, 0] = (b[:, 0] & np.uint8((0xFF << (mode + 1)) & 0xFF)) | np.uint8(1 << mode)
            out.append(("mode%d_extremes" % mode, b))
    elif name in ("BPTC_FLOAT", "BPTC_SIGNED_FLOAT"):
        for mode, code in enumerate(BC6H_MODE_CODES):
            b = R("mode%d" % mode)
            b[:, 0] = (b[:, 0] & np.uint8(0xFC if mode < 2 else 0xE0)) | np.uint8(code)
            out.append(("mode%d" % mode, b))
            e = np.where(np.arange(n)[:, None] % 2, 0xFF, 0x00).astype(np.uint8) * np.ones((n, bs), np.uint8)
            e[:, 10:] = R("ext%d" % mode)[:, 10:]
            e[:, 0] = (e[:, 0] & np.uint8(0xFC if mode < 2 else 0xE0)) | np.uint8(code)
            out.append(("mode%d_extremes" % mode, e))
        for code in BC6H_RESERVED:
            b = R("res%02x" % code); b[:, 0] = (b[:, 0] & np.uint8(0xE0)) | np.uint8(code); out.append(("reserved_%02x" % code, b))
    zeros = np.zeros((4, bs), np.uint8); ones = np.full((4, bs), 0xFF, np.uint8)
    out += [("all_zero", zeros), ("all_ones", ones)]
    return [(lab, np.ascontiguousarray(b)) for lab, b in out]


def forced_stream(fmt):
    """all classes concatenated -> (blocks[n,bs], labels per block)"""
    cls = forced_classes(fmt)
    blocks = np.concatenate([b for _, b in cls])
    labels = sum(([lab] * len(b) for lab, b in cls), [])
    return blocks, labels


def stream_m(fmt, base):
    """Stream M: stream U with the mode field overwritten round-robin so that all
    valid modes of BPTC / BPTC_FLOAT are equiprobable; other formats are returned unchanged."""
    b = np.array(base, dtype=np.uint8).reshape(-1, fmt.block_bytes)
    i = np.arange(len(b))
    if fmt.name == "BPTC":
        m = (i % 8).astype(np.uint8)
        b[:, 0] = (b[:, 0] & (0xFF << (m + 1)).astype(np.uint8)) | (1 << m).astype(np.uint8)
    elif fmt.name in ("BPTC_FLOAT", "BPTC_SIGNED_FLOAT"):
        codes = np.array(BC6H_MODE_CODES, np.uint8)[i % 14]
        keep = np.where((i % 14) < 2, 0xFC, 0xE0).astype(np.uint8)
        b[:, 0] = (b[:, 0] & keep) | codes
    return b.reshape(-1)


def stream_c(fmt, wb, hb):
    """Stream C: the reference's bundled 64x64 fixture of this format (16x16 blocks,
    tests/golden/test-texture-<FMT>.ktx; file list validate.c:31-57) tiled over wb x hb blocks --
    coherent, encoder-made content.  None for the two formats the reference ships no fixture for."""
    import os
    from detex_amd.ktx import read_ktx
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "test-texture-%s.ktx" % fmt.name)
    if not os.path.exists(path):
        return None
    k = read_ktx(path)
    fw, fh = k["width_in_blocks"], k["height_in_blocks"]
    tile = k["data"].reshape(fh, fw, fmt.block_bytes)
    reps = ((hb + fh - 1) // fh, (wb + fw - 1) // fw, 1)
    return np.ascontiguousarray(np.tile(tile, reps)[:hb, :wb]).reshape(-1)


def make_stream(kind, fmt, wb, hb, seed=None):
    """'U', 'M' or 'C' stream of wb x hb blocks (None if the kind does not exist for fmt).  Measurement-only kinds: 'Z' all-zero
    blocks (every pixel decodes to zero for the formats it is used with), 'S' stream C with the fixture's blocks shuffled (same
    blocks, same modes, no spatial coherence), 'K' one block of stream U repeated (constant image), 'F' the unsigned BC6H fixture as signed BC6H blocks, 'm0'..'m7' stream U with every
    BPTC block forced to that mode (uniform waves, random content)."""
    import oracle_lib as ol
    if len(kind) == 2 and kind[0] == "m":
        if fmt.name != "BPTC":
            return None
        b = np.array(ol.stream_u(fmt, wb * hb, seed=seed), dtype=np.uint8).reshape(-1, fmt.block_bytes)
        m = int(kind[1])
        b[:, 0] = (b[:, 0] & np.uint8((0xFF << (m + 1)) & 0xFF)) | np.uint8(1 << m)
        return b.reshape(-1)
    if kind == "C":
        return stream_c(fmt, wb, hb)
    if kind == "F":     # measurement only: signed BC6H fed the unsigned format's fixture (the same block syntax; coherent content)
        if fmt.name != "BPTC_SIGNED_FLOAT":
            return None
        return stream_c(F.BY_NAME["BPTC_FLOAT"], wb, hb)
    if kind == "Z":
        return np.zeros(wb * hb * fmt.block_bytes, np.uint8)
    if kind == "S":
        c = stream_c(fmt, wb, hb)
        if c is None:
            return None
        b = c.reshape(-1, fmt.block_bytes)
        return np.ascontiguousarray(b[np.random.default_rng(7).permutation(len(b))]).reshape(-1)
    if kind == "K":
        one = stream_m(fmt, ol.stream_u(fmt, 64, seed=seed)).reshape(-1, fmt.block_bytes)[5]
        return np.ascontiguousarray(np.tile(one, wb * hb))
    data = ol.stream_u(fmt, wb * hb, seed=seed)
    return stream_m(fmt, data) if kind == "M" else data
