// decode_bptc.h -- BPTC (BC7), all eight modes, one lane per block, gfx950.
//
// The reference decodes with an 8-way mode switch, a bit-at-a-time 128-bit reader and per-texel table lookups
// (decompress-bptc.c:354-512).  A wavefront of 64 independent blocks would execute every taken mode path
// serially, so this decoder is ONE divergence-free data-driven path whose VALU work is kept as small as the
// format allows (the kernel is VALU-issue-bound, not HBM-bound: DESIGN.md section 4):
//   * everything about a mode that does not depend on the block -- field positions as (LDS row, shift) pairs,
//     widths, packed per-lane shift amounts of the 8-bit expansion, P-bit routing, index widths, masks and
//     weight constants of the colour and alpha index streams -- is a Bc7Rec derived AT COMPILE TIME from the
//     eight mode descriptors and fetched per lane from a workgroup LDS copy (record 8 = mode 4 with the index
//     selection bit set, so stream swapping costs nothing);
//   * the block's bits live in per-lane LDS rows: a field is one address add, one ds_read2st64_b32 and one
//     v_alignbit_b32;
//   * endpoints are gathered straight into the 16-bit lanes the blend needs, (R,G) and (B,A), and expanded to
//     8 bits there with per-lane packed shifts (v_pk_lshlrev_b16 / v_pk_lshrrev_b16) and v_bitop3_b32;
//   * the anchor texels' missing index bits are INSERTED (w += w & himask: two full-rate ops per anchor, masks
//     from a compile-time-derived table indexed by partition) so every texel reads its index at a regular
//     position: `and` + `shift`, both full-rate, instead of per-texel widths;
//   * a weight is one v_mad_u32_u24 whose high half feeds both lanes of v_pk_mad_u16 through op_sel (no
//     broadcast instruction); the (up to three) subsets' blend operands sit in per-lane LDS rows at a 16 KiB
//     aligned base, so a texel's subset row address is one v_bitop3_b32 of the pre-shifted partition word.
// Partition / anchor constants are the bit-packed words of bptc_tables.inc.
//
// Waves whose 64 blocks share one record (encoder output is mode-coherent; the uniform-random stream never
// is) branch to a copy of the same code with the record as a compile-time constant (DecBPTC; kUniform).
//
// Reference quirk reproduced: in mode 6 the second P-bit (block bit 64) reads 0.
#pragma once
#include <cstddef>
#include "bptc_common.h"
#include "decode_s3tc_rgtc.h"

namespace detexhip {

// per-mode layout (decompress-bptc.c:24-43 comment table, :45-71, :134, :195-225, :265-267)
struct Bc7ModeDesc { uint32_t ns, pb, rb, isb, cb, ab, epb, spb, ib, ib2; };
constexpr Bc7ModeDesc kBc7Modes[8] = {
	//subsets part rot isel colour alpha endpoint-P shared-P index index2
	{ 3, 4, 0, 0, 4, 0, 1, 0, 3, 0 }, { 2, 6, 0, 0, 6, 0, 0, 1, 3, 0 }, { 3, 6, 0, 0, 5, 0, 0, 0, 2, 0 }, { 2, 6, 0, 0, 7, 0, 1, 0, 2, 0 },
	{ 1, 0, 2, 1, 5, 6, 0, 0, 2, 3 }, { 1, 0, 2, 0, 7, 8, 0, 0, 2, 2 }, { 1, 0, 0, 0, 7, 7, 1, 0, 4, 0 }, { 2, 6, 0, 0, 5, 5, 1, 0, 2, 0 } };

// ---- partition / anchor table -------------------------------------------------------------------------
// One entry per (partition table, index width) pair a block can name, 2 dwords:
//   pword   2-bit subset number per texel
//   route   five 5-bit fields: positions of the zero bits to insert into the first index window (texels 0-7) for the
//           anchors of subsets 1 and 2 that lie there, ascending (lo1, lo2), the same for the second window (texels
//           8-15: hi1, hi2) -- 31 = none: every window of a partitioned mode is at most 24 bits wide, so an insertion
//           at bit 31 touches nothing that is read (one-subset modes shift a zero word instead of all-ones: record
//           field ins_ones) -- and `half`, the bits the first window consumes from the stream
//           = 8*ib - (anchors among texels 0-7).  Texel 0's own anchor bit is in the record.
// Sections: [0,64) two subsets ib 2 (modes 3, 7) . [64,128) two subsets ib 3 (mode 1) . [128,192) three
// subsets ib 2 (mode 2) . [192,208) three subsets ib 3 (mode 0, 4-bit partition number) . 208/209/210 one
// subset with ib 2/3/4 (modes 4, 5, 6: only `half` matters).
constexpr int kBc7PartEntries = 211;
struct Bc7PartEntry { uint32_t pword, route; };
struct Bc7PartTable { Bc7PartEntry e[kBc7PartEntries]; };
constexpr Bc7PartEntry bc7_part_entry(int subsets, uint32_t ib, uint32_t part) {
	Bc7PartEntry r = {};
	uint32_t a[2] = { 99u, 99u };
	if (subsets == 2) { a[0] = kAnchorWordsCx[part] & 15u; r.pword = kPartition2BitCx[part]; }
	if (subsets == 3) {
		a[0] = (kAnchorWordsCx[part] >> 4) & 15u; a[1] = (kAnchorWordsCx[part] >> 8) & 15u;
		r.pword = kPartition2BitCx[64u + part];
		if (a[0] > a[1]) { const uint32_t t = a[0]; a[0] = a[1]; a[1] = t; }
	}
	uint32_t nlo = 1u, lo[2] = { 31u, 31u }, hi[2] = { 31u, 31u }, klo = 0u, khi = 0u;
	for (int k = 0; k < 2; k++) {
		if (a[k] > 15u) continue;
		const uint32_t pos = (a[k] & 7u) * ib + ib - 1u;	// where the anchor's absent MSB belongs in its window
		if (a[k] < 8u) { lo[klo++] = pos; nlo++; } else hi[khi++] = pos;
	}
	r.route = lo[0] | (lo[1] << 5) | (hi[0] << 10) | (hi[1] << 15) | ((8u * ib - nlo) << 20);
	return r;
}
constexpr Bc7PartTable bc7_part_table() {
	Bc7PartTable t = {};
	for (uint32_t p = 0; p < 64u; p++) {
		t.e[p] = bc7_part_entry(2, 2u, p);
		t.e[64u + p] = bc7_part_entry(2, 3u, p);
		t.e[128u + p] = bc7_part_entry(3, 2u, p);
		if (p < 16u) t.e[192u + p] = bc7_part_entry(3, 3u, p);
	}
	t.e[208] = bc7_part_entry(1, 2u, 0u); t.e[209] = bc7_part_entry(1, 3u, 0u); t.e[210] = bc7_part_entry(1, 4u, 0u);
	return t;
}
__constant__ Bc7PartTable kBc7PartTable = bc7_part_table();

// ---- per-mode record ------------------------------------------------------------------------------------
// 52 words in thirteen 16-byte groups, ordered by use: the decoder fetches a group with ONE ds_read_b128 (4 LDS cycles
// for the wave) at the point where its fields are needed.  Left to field-wise reads the compiler issued ds_read_b96 (8
// cycles) and ds_read2_b32 (4 cycles per 8 bytes) for the same bytes: about 40 of the 250 LDS cycles a wave spends per tile.
constexpr uint32_t kBc7RowBytes = 1024u;	// LDS stride between a lane's consecutive block dwords ([row][lane], 256 lanes)
enum Bc7Field : int {
	F_POS_PART, F_PB, F_POS_ROT, F_RB,			// G0  header fields of the first dword
	F_PART_BASE, F_NS, F_MODE, F_TWO,			// G1  BYTE offset of the mode's section of the partition table; subsets; mode; has a second index stream
	F_ROW_R, F_ROW_G, F_ROW_B, F_ROW_A,			// G2  LDS row byte offset of the dword a channel's fields start in
	F_POS_R, F_POS_G, F_POS_B, F_POS_A,			// G3  their bit positions (v_alignbit_b32 uses the low 5 bits)
	F_ROW_P, F_POS_P, F_P_WORD_MASK, F_CB,			// G4  P-bits; QUIRK A-2 mask; colour bits
	F_AB, F_UP_C, F_DOWN_C, F_PCONST_RG,			// G5  alpha bits; colour shift-up 8 - cb; packed per-16-bit-lane shift amounts; P-bit place in (R,G)
	F_UP_BA, F_DOWN_BA, F_PCONST_BA, F_SET_BA,		// G6  the same for (B,A); alpha = 255 for modes 0-3
	F_PIDX0, F_PIDX1, F_PIDX2, F_PIDX3,			// G7  P-bit of endpoint e
	F_PIDX4, F_PIDX5, F_INS_ONES, F_HIMASK0_C,		// G8  ~0 if the mode has partitions; texel 0's anchor bit (colour stream)
	F_ROW_C, F_POS_C, F_IBC, F_IMASK_C,			// G9  colour index stream
	F_WMUL_C, F_WADD_C, F_SEL_COMB, F_HALF_A,		// G10 weight = byte 2 of index * mul + add; v_perm selector building {colour, alpha} quarter windows
	F_ROW_A2, F_POS_A2, F_IB4_C, F_IB4_A,			// G11 alpha index stream (modes 4, 5); bits per four texels of either stream
	F_APAIR, F_IMASK_PAIR, F_HIMASK0_A, F_IB_PAIR,		// G12 the two streams side by side in 16-bit lanes: weight multiplier, index mask, index width
	kBc7RecWords
};
struct alignas(16) Bc7Rec { uint32_t w[kBc7RecWords]; };
static_assert(sizeof(Bc7Rec) == 208, "record is fetched with 16-byte LDS reads; 52-dword stride keeps records on disjoint banks");
constexpr int kBc7Recs = 9;

constexpr Bc7Rec bc7_rec(uint32_t mode, bool isel) {
	const Bc7ModeDesc m = kBc7Modes[mode];
	Bc7Rec L = {};
	L.w[F_MODE] = mode;
	L.w[F_POS_PART] = mode + 1u; L.w[F_PB] = m.pb;
	L.w[F_POS_ROT] = mode + 1u + m.pb; L.w[F_RB] = m.rb;
	const uint32_t pos_isel = mode + 1u + m.pb + m.rb;
	const uint32_t chan = 2u * m.ns * m.cb;
	const uint32_t pos_r = pos_isel + m.isb, pos_g = pos_r + chan, pos_b = pos_g + chan, pos_a = pos_b + chan;
	const uint32_t pos_p = pos_a + 2u * m.ns * m.ab;
	const uint32_t pos_idx = pos_p + m.epb * 2u * m.ns + m.spb * m.ns;
	const uint32_t pos_idx2 = pos_idx + 16u * m.ib - m.ns;
	L.w[F_CB] = m.cb; L.w[F_AB] = m.ab; L.w[F_NS] = m.ns;
	L.w[F_PART_BASE] = (uint32_t)sizeof(Bc7PartEntry) *
		(m.ns == 1u ? 208u + ((isel ? m.ib2 : m.ib) - 2u) : (m.ns == 2u ? (m.ib == 2u ? 0u : 64u) : (m.ib == 2u ? 128u : 192u)));
	L.w[F_ROW_R] = (pos_r >> 5) * kBc7RowBytes; L.w[F_ROW_G] = (pos_g >> 5) * kBc7RowBytes;
	L.w[F_ROW_B] = (pos_b >> 5) * kBc7RowBytes; L.w[F_ROW_A] = (pos_a >> 5) * kBc7RowBytes;
	L.w[F_POS_R] = pos_r; L.w[F_POS_G] = pos_g; L.w[F_POS_B] = pos_b; L.w[F_POS_A] = pos_a;
	L.w[F_ROW_P] = (pos_p >> 5) * kBc7RowBytes; L.w[F_POS_P] = pos_p;
	L.w[F_P_WORD_MASK] = mode == 6u ? 1u : 0xFFFFFFFFu;			// QUIRK A-2 (decompress-bptc.c:142-146)
	// 8-bit expansion of a cb-bit value v with optional P-bit p (decompress-bptc.c:136-180):
	//   (v << (8 - cb)) | (p << (7 - cb)) | (v >> (cb + cprec - 8)),  cprec = cb + has_p
	const uint32_t has_p = m.epb | m.spb, cprec = m.cb + has_p, aprec = m.ab + m.epb;
	const uint32_t up_c = 8u - m.cb, down_c = m.cb + cprec - 8u;
	const uint32_t up_a = m.ab ? 8u - m.ab : 0u, down_a = m.ab ? m.ab + aprec - 8u : 0u;
	L.w[F_UP_C] = up_c;
	L.w[F_DOWN_C] = down_c * 0x00010001u;
	L.w[F_UP_BA] = up_c | (up_a << 16);
	L.w[F_DOWN_BA] = down_c | (down_a << 16);
	const uint32_t pc = has_p ? 1u << (7u - m.cb) : 0u, pa = (m.ab && m.epb) ? 1u << (7u - m.ab) : 0u;
	L.w[F_PCONST_RG] = pc * 0x00010001u;
	L.w[F_PCONST_BA] = pc | (pa << 16);
	L.w[F_SET_BA] = m.ab ? 0u : 0x00FF0000u;				// modes 0-3 are opaque (:176-179)
	// P-bit of endpoint e: its own (epb), its subset's (spb, mode 1), none (pconst = 0)
	const uint32_t sh = m.spb ? 1u : 0u;
	for (uint32_t e = 0; e < 4u; e++) L.w[F_PIDX0 + e] = e >> sh;
	L.w[F_PIDX4] = 4u >> sh; L.w[F_PIDX5] = 5u >> sh;
	// index streams (:401-480): primary 16*ib - ns bits, then (modes 4/5) the secondary one; the colour stream is
	// the primary one unless the index-selection bit swaps them (:374-375, 452-480)
	const bool two = m.ib2 != 0u;
	const uint32_t ibc = (two && isel) ? m.ib2 : m.ib, iba = two ? (isel ? m.ib : m.ib2) : m.ib;
	const uint32_t pos_c = (two && isel) ? pos_idx2 : pos_idx, pos_a2 = two ? (isel ? pos_idx : pos_idx2) : 0u;
	L.w[F_ROW_C] = (pos_c >> 5) * kBc7RowBytes; L.w[F_POS_C] = pos_c; L.w[F_IBC] = ibc; L.w[F_IMASK_C] = (1u << ibc) - 1u;
	L.w[F_WMUL_C] = bptc_weight_mul(ibc); L.w[F_WADD_C] = bptc_weight_add(ibc);
	L.w[F_HIMASK0_C] = 0xFFFFFFFFu << (ibc - 1u);
	L.w[F_TWO] = two ? 1u : 0u;
	L.w[F_ROW_A2] = (pos_a2 >> 5) * kBc7RowBytes; L.w[F_POS_A2] = pos_a2;
	L.w[F_HIMASK0_A] = 0xFFFFFFFFu << (iba - 1u);
	L.w[F_HALF_A] = 8u * iba - 1u;
	// Waves that hold a two-stream block (kernel path `any_two`) walk BOTH streams in lockstep, four texels at a time: a quarter
	// window is {colour indices, alpha indices} in the two 16-bit lanes of one register (4 texels x <= 4 bits each), masked, turned
	// into the weight pair and advanced by ONE packed instruction each.  Blocks without a second stream carry the colour stream in
	// both lanes (selector, multiplier, mask and width of the colour stream twice).
	L.w[F_SEL_COMB] = two ? 0x05040100u : 0x01000100u;			// v_perm(alpha window, colour window, .): low halves side by side
	L.w[F_IB4_C] = 4u * ibc; L.w[F_IB4_A] = 4u * iba;
	L.w[F_APAIR] = bptc_weight16_mul(ibc) | (bptc_weight16_mul(two ? iba : ibc) << 16);
	L.w[F_IMASK_PAIR] = ((1u << ibc) - 1u) | (((1u << (two ? iba : ibc)) - 1u) << 16);
	L.w[F_IB_PAIR] = ibc | ((two ? iba : ibc) << 16);
	L.w[F_INS_ONES] = m.ns == 1u ? 0u : 0xFFFFFFFFu;
	return L;
}
struct Bc7RecTable { Bc7Rec r[kBc7Recs]; };
constexpr Bc7RecTable bc7_rec_table() {
	Bc7RecTable t = {};
	for (uint32_t m = 0; m < 8u; m++) t.r[m] = bc7_rec(m, false);
	t.r[8] = bc7_rec(4u, true);
	return t;
}
__constant__ Bc7RecTable kBc7RecTable = bc7_rec_table();
constexpr Bc7RecTable kBc7RecTableCx = bc7_rec_table();

// v_perm_b32 selectors gathering the high bytes of the four 16-bit sums (R byte 1, G 3, B 5, A 7), with the
// mode 4/5 rotation that swaps A with R/G/B (:497-508)
__constant__ uint32_t kBc7Gather[4] = { 0x07050301u, 0x01050307u, 0x03050701u, 0x05070301u };

// ---- LDS: one struct at a 16 KiB-aligned address -----------------------------------------------------------------------
// subset rows first (row s of lane l at 4096*s + 16*l: bits 12-13 of a lane's base are clear, so the row address
// of a texel is bitop3((pword pre-shifted) & 0x3000 | base)); the fourth 4 KiB slot and what follows hold the
// workgroup tables and the per-lane block dwords.
// the workgroup tables as ONE constant image, so that the copy into LDS is a single 16-byte load and a single
// ds_write_b128 per thread (224 of the 256 threads) instead of four dword loads and stores each
struct alignas(16) Bc7Tables {
	Bc7Rec rec[kBc7Recs];
	uint32_t gather[4];
	Bc7PartEntry part[kBc7PartEntries];
	uint32_t pad[2];
};
static_assert(sizeof(Bc7Tables) % 16 == 0 && sizeof(Bc7Tables) / 16 <= 256, "one 16-byte vector per thread");
constexpr Bc7Tables bc7_tables() {
	Bc7Tables t = {};
	const Bc7RecTable r = bc7_rec_table();
	const Bc7PartTable p = bc7_part_table();
	for (int k = 0; k < kBc7Recs; k++) t.rec[k] = r.r[k];
	for (int k = 0; k < kBc7PartEntries; k++) t.part[k] = p.e[k];
	t.gather[0] = 0x07050301u; t.gather[1] = 0x01050307u; t.gather[2] = 0x03050701u; t.gather[3] = 0x05070301u;	// = kBc7Gather
	return t;
}
__constant__ Bc7Tables kBc7Tables = bc7_tables();

// ---- storage: the one part of this decoder that differs between the device and the emulation of tests/host_emul -------------------
#if defined(__HIPCC__)
struct Bc7Lds {
	uint4 subset[3][256];			// per-lane blend operands {base_rg, base_ba, 4*diff_rg, 4*diff_ba}
	Bc7Tables t;
	// per-lane block dwords, LAST member: fields that start near the end of the block also read "rows" 4 and 5,
	// i.e. up to 2 KiB past this array -- LDS reads beyond the allocation return 0 rather than faulting, and whatever
	// they return stands for bits beyond 127, which no field or index ever consumes
	alignas(16) uint32_t bits[4][256];	// (16-byte aligned: the block-major exchange stages 16-byte vectors here, stage_slot)
};
static_assert(sizeof(Bc7Lds) <= 20480, "eight workgroups per CU by LDS (the round-2 occupancy sweep padded this struct: profiles/AB_RECORD.md)");
static_assert(offsetof(Bc7Lds, bits) + sizeof(Bc7Lds::bits) == sizeof(Bc7Lds), "bits[] must stay the LAST member: fields near the end of a block read up to two rows past it (see the comment in the struct)");
DH Bc7Lds &bc7_lds() { __shared__ __attribute__((aligned(16384))) Bc7Lds s; return s; }	// the VARIABLE is aligned: the size is not rounded up
// (Requesting the kernel's first block between the table load and its LDS store -- so that the block travels during the
// barrier -- was measured too: the compiler issues the block load first either way, and then the barrier waits for the
// slowest wave's HBM round trip: stream C 49.4 vs 48.3 us.  The block is loaded after the barrier.)
DH void bc7_prepare() {
	typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
	if (threadIdx.x < sizeof(Bc7Tables) / 16u)
		reinterpret_cast<u32x4 *>(&bc7_lds().t)[threadIdx.x] = reinterpret_cast<const u32x4 *>(&kBc7Tables)[threadIdx.x];
	__syncthreads();
}
// 16-byte staging slot of the block-major exchange (kernels.h: decode_blocks) inside this wave's own lane rows, which
// are dead once a tile is decoded: vector k (0..3) of the wave's block b (0..63).  Vectors 0-2 live in the wave's
// 1 KiB of subset row k, vector 3 in the wave's four 256-byte pieces of the block-dword rows; the slot number is b with
// 4k XORed in, so that the transposed reads (four consecutive lanes = the four vectors of one block, sixteen lanes = four
// blocks) fall on sixty-four different banks -- and, XOR touching only bits 2-3 of b, a lane's slots for blocks b, b + 16,
// b + 32, b + 48 are 256 bytes (vector 3: 1 KiB) apart: one address per lane, the rest immediate offsets (round 3 rotated by
// 2k modulo 64: an address computation per read, ~60 VALU per wave in this exchange).  A separate 17 KiB staging array left
// four workgroups per CU resident (block-major BC7: 65 us against 58 linear).
DH void *bc7_stage_slot(uint32_t k, uint32_t b) {
	Bc7Lds &s = bc7_lds();
	const uint32_t w = threadIdx.x >> 6, p = b ^ (4u * k);
	char *in_rows = reinterpret_cast<char *>(&s.subset[0][64u * w]) + k * (uint32_t)sizeof(s.subset[0]) + p * 16u;
	char *in_bits = reinterpret_cast<char *>(&s.bits[0][64u * w]) + (p >> 4) * (uint32_t)sizeof(s.bits[0]) + (p & 15u) * 16u;
	return k < 3u ? in_rows : in_bits;
}
// bytes from stage_slot(k, b) to stage_slot(k, b + 16) (b < 48; the XOR leaves bits 4-5 of b alone)
DH uint32_t bc7_stage_step(uint32_t k) { return k < 3u ? 256u : (uint32_t)sizeof(bc7_lds().bits[0]); }

// the per-lane view of that storage
struct Bc7Lane {
	uint32_t bits_base, subset_base;	// LDS byte addresses of this lane's column
	uint32_t subset_bits;			// 0x3000 in a VGPR (see get_subset)
	DH Bc7Lane() {
		Bc7Lds &s = bc7_lds();
		bits_base = (uint32_t)(uintptr_t)&s.bits[0][threadIdx.x];
		subset_base = (uint32_t)(uintptr_t)&s.subset[0][threadIdx.x];
		subset_bits = 0x3000u;
		if constexpr (Tune::kMasksInVgprs) pin_vgpr(subset_bits);	// (v_bitop3_b32 takes no literal: a VGPR, not an SGPR source at half rate)
	}
	typedef __attribute__((address_space(3))) uint32_t lds_u32;
	typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
	typedef __attribute__((address_space(3))) u32x4 lds_u4;
	DH void put_bits(uint4 blk) const {
		lds_u32 *p = (lds_u32 *)(uintptr_t)bits_base;
		p[0] = blk.x; p[256] = blk.y; p[512] = blk.z; p[768] = blk.w;
	}
	// bits [pos, pos+32) of the block; row = (pos >> 5) * kBc7RowBytes
	DH uint32_t field(uint32_t row, uint32_t pos) const {
		const lds_u32 *p = (const lds_u32 *)(uintptr_t)(bits_base + row);
		return __builtin_amdgcn_alignbit(p[256], p[0], pos);
	}
	// bits [pos, pos+32) and [pos+32, pos+64)
	DH void field64(uint32_t row, uint32_t pos, uint32_t &lo, uint32_t &hi) const {
		const lds_u32 *p = (const lds_u32 *)(uintptr_t)(bits_base + row);
		const uint32_t d0 = p[0], d1 = p[256], d2 = p[512];
		lo = __builtin_amdgcn_alignbit(d1, d0, pos);
		hi = __builtin_amdgcn_alignbit(d2, d1, pos);
	}
	DH void put_subset(int s, uint4 v) const { ((lds_u4 *)(uintptr_t)subset_base)[s * 256] = u32x4{ v.x, v.y, v.z, v.w }; }
	// sel: any word with the subset number at bits 12-13
	DH uint4 get_subset(uint32_t sel) const {
		const u32x4 v = *(const lds_u4 *)(uintptr_t)(uint32_t)__builtin_amdgcn_bitop3_b32(sel, subset_bits, subset_base, 0xEA);
		return uint4{ v.x, v.y, v.z, v.w };
	}
	// byte address of record r in the workgroup's LDS image, opaque to the optimiser: the groups then come as
	// immediate offsets from one address register
	static DH uint32_t rec_address(uint32_t r) {
		return opaque((uint32_t)(uintptr_t)bc7_lds().t.rec + r * (uint32_t)sizeof(Bc7Rec));
	}
	// group G (four consecutive words) of the record at `address`: ONE ds_read_b128.  pin() keeps such reads whole (the
	// compiler otherwise narrows a read to ds_read_b96 / ds_read2_b32 when a word is unused: 8 and 4 LDS cycles per wave
	// instead of 4) and is where the wave waits for them -- so a batch of groups is requested first and pinned together.
	typedef u32x4 Group;
	template <int G> static DH Group rec_group(uint32_t address) { return *(const lds_u4 *)(uintptr_t)(address + 16u * G); }
	static DH void pin(Group &a) { pin_vgpr(a); }
	static DH void pin(Group &a, Group &b) { pin_vgpr(a, b); }
	static DH void pin(Group &a, Group &b, Group &c, Group &d, Group &e) { pin_vgpr(a, b, c, d, e); }
	static DH const Bc7PartEntry &part(uint32_t byte_offset) {
		return *reinterpret_cast<const Bc7PartEntry *>(reinterpret_cast<const char *>(bc7_lds().t.part) + byte_offset);
	}
	static DH uint32_t gather(uint32_t rot) { return bc7_lds().t.gather[rot]; }
};
#else
// THE SAME INTERFACE FOR THE GPU-LESS EMULATION (tests/host_emul): LDS byte addresses held in 32-bit registers do not exist on a
// host, so the per-lane storage is plain arrays and the workgroup tables are read where they are; everything above and below
// this block is the code the device runs.
DH void bc7_prepare() {}
DH void *bc7_stage_slot(uint32_t, uint32_t) { return nullptr; }
DH uint32_t bc7_stage_step(uint32_t) { return 0u; }
struct Bc7Lane {
	uint32_t bits[6];
	uint4 subset[3];
	DH Bc7Lane() { for (int k = 0; k < 6; k++) bits[k] = 0xA5A5A5A5u; }	// rows 4, 5: arbitrary (device: whatever follows in LDS)
	DH void put_bits(uint4 blk) { bits[0] = blk.x; bits[1] = blk.y; bits[2] = blk.z; bits[3] = blk.w; }
	DH uint32_t field(uint32_t row, uint32_t pos) const {
		const uint32_t k = row / kBc7RowBytes;
		return __builtin_amdgcn_alignbit(bits[k + 1], bits[k], pos);
	}
	DH void field64(uint32_t row, uint32_t pos, uint32_t &lo, uint32_t &hi) const {
		const uint32_t k = row / kBc7RowBytes;
		lo = __builtin_amdgcn_alignbit(bits[k + 1], bits[k], pos);
		hi = __builtin_amdgcn_alignbit(bits[k + 2], bits[k + 1], pos);
	}
	DH void put_subset(int s, uint4 v) { subset[s] = v; }
	DH uint4 get_subset(uint32_t sel) const { return subset[(sel >> 12) & 3u]; }
	static DH uint32_t rec_address(uint32_t r) { return r; }
	typedef uint4 Group;
	template <int G> static DH Group rec_group(uint32_t r) {
		const uint32_t *w = kBc7RecTable.r[r].w + 4 * G;
		return uint4{ w[0], w[1], w[2], w[3] };
	}
	static DH void pin(Group &) {}
	static DH void pin(Group &, Group &) {}
	static DH void pin(Group &, Group &, Group &, Group &, Group &) {}
	static DH const Bc7PartEntry &part(uint32_t byte_offset) { return kBc7PartTable.e[byte_offset / sizeof(Bc7PartEntry)]; }
	static DH uint32_t gather(uint32_t rot) { return kBc7Gather[rot]; }
};
#endif

// FIXED >= 0: the record is a compile-time constant (every lane of the wave is known to use record FIXED)
template <int FIXED> struct Bc7RecReader {
	typedef Bc7Lane::Group Group;
	uint32_t address;
	DH explicit Bc7RecReader(uint32_t rec_index) : address(FIXED >= 0 ? 0u : Bc7Lane::rec_address(rec_index)) {}
	template <int G> DH Group group() const {
		if constexpr (FIXED >= 0) {
			constexpr Bc7Rec kFixed = kBc7RecTableCx.r[FIXED >= 0 ? FIXED : 0];
			return Group{ kFixed.w[4 * G], kFixed.w[4 * G + 1], kFixed.w[4 * G + 2], kFixed.w[4 * G + 3] };
		} else {
			return Bc7Lane::rec_group<G>(address);
		}
	}
	// the wave waits here for the groups named (a no-op for a compile-time record)
	template <class... Gs> DH void pin(Gs &...gs) const { if constexpr (FIXED < 0) Bc7Lane::pin(gs...); }
};

template <int FIXED>
DH void bc7_decode_with(uint4 blk, uint32_t rec_index, uint32_t flags, uint32_t (&d)[16]) {
	constexpr Bc7Rec kFixed = kBc7RecTableCx.r[FIXED >= 0 ? FIXED : 0];
	const Bc7RecReader<FIXED> L(rec_index);
	typedef Bc7Lane::Group Group;
	Group g_head = L.template group<F_POS_PART / 4>();	// pos_part, pb, pos_rot, rb
	Group g_part = L.template group<F_PART_BASE / 4>();	// part_base, ns, mode, two
	// requested now, waited for where the endpoint phase needs them
	Group g_row = L.template group<F_ROW_R / 4>(), g_pos = L.template group<F_POS_R / 4>();
	Group g_p = L.template group<F_ROW_P / 4>();		// row_p, pos_p, p_word_mask, cb
	Group g_c = L.template group<F_AB / 4>();		// ab, up_c, down_c, pconst_rg
	Group g_ba = L.template group<F_UP_BA / 4>();		// up_ba, down_ba, pconst_ba, set_ba
	Group g_pi = L.template group<F_PIDX0 / 4>();		// pidx0..3
	Group g_pj = L.template group<F_PIDX4 / 4>();		// pidx4, pidx5, ins_ones, himask0_c
	L.pin(g_head, g_part);
	const uint32_t mode = g_part.z;
	Bc7Lane lane;
	lane.put_bits(blk);
	stage_priority<Tune::kBc7Prio, 0>();

	// header fields all lie in the first 14 bits
	const uint32_t part = ubfe(blk.x, g_head.x, g_head.y);
	const uint32_t rot = ubfe(blk.x, g_head.z, g_head.w);
	const Bc7PartEntry &pe = Bc7Lane::part(DETEX_UMUL24(part, (uint32_t)sizeof(Bc7PartEntry)) + g_part.x);
	const uint32_t gather = Bc7Lane::gather(rot);

	// wave-uniform trimming: endpoints of subsets no lane of the wave has are not expanded, alpha fields are
	// skipped in waves of opaque modes, the second index stream in waves without modes 4/5 (the uniform-random
	// stream has all of them in nearly every wave; encoder output mostly does not)
	const uint32_t wave_subsets = FIXED >= 0 ? kFixed.w[F_NS]
		: (__builtin_amdgcn_ballot_w64(g_part.y == 3u) ? 3u : (__builtin_amdgcn_ballot_w64(g_part.y == 2u) ? 2u : 1u));
	const bool wave_alpha = FIXED >= 0 ? kFixed.w[F_AB] != 0u : __builtin_amdgcn_ballot_w64(mode >= 4u) != 0;
	const bool any_two = FIXED >= 0 ? kFixed.w[F_TWO] != 0u : __builtin_amdgcn_ballot_w64(g_part.w != 0u) != 0;

	// endpoint fields: all R, then all G, then all B, then all A (each 2*ns values), then the P-bits (:74-132)
	L.pin(g_row, g_pos);
	const uint32_t wr = lane.field(g_row.x, g_pos.x), wg = lane.field(g_row.y, g_pos.y), wb = lane.field(g_row.z, g_pos.z);
	const uint32_t wa = wave_alpha ? lane.field(g_row.w, g_pos.w) : 0u;
	L.pin(g_p, g_c, g_ba, g_pi, g_pj);
	// QUIRK A-2 lives in the record's P-word mask (mode 6: only the first P-bit survives); the spec switch, wave-uniform, lifts it
	const uint32_t pw = lane.field(g_p.x, g_p.y) & (g_p.z | ((flags & kFlagSpecBc7Mode6PBit) ? 0xFFFFFFFFu : 0u));
	const uint32_t cb = g_p.w, ab = g_c.x;
	const uint32_t pidx[6] = { g_pi.x, g_pi.y, g_pi.z, g_pi.w, g_pj.x, g_pj.y };
	uint32_t off = 0u, offa = 0u;
	// With e0, e1 in 0..255 and w in 0..64 the reference's ((64-w)*e0 + w*e1 + 32) >> 6 (:182-193) equals the
	// high byte of 256*e0 + 128 + 4*w*(e1 - e0) (range 128 .. 65408: fits a 16-bit lane, exact mod 2^16), so a
	// texel is two v_pk_mad_u16 and one v_perm_b32 that gathers the four high bytes.  Per subset: its two endpoints
	// expanded, then the row {base_rg, base_ba, 4*diff_rg, 4*diff_ba} parked in the lane's LDS column (nothing of a
	// subset the wave does not have is computed, initialised or kept).
#pragma unroll
	for (int s = 0; s < 3; s++) {
		if ((uint32_t)s >= wave_subsets) break;
		uint32_t x_rg[2], x_ba[2];
#pragma unroll
		for (int k = 0; k < 2; k++) {
			const int e = 2 * s + k;
			// straight into the blend's 16-bit lanes: (R, G) and (B, A)
			const uint32_t rg = (ubfe(wg, off, cb) << 16) | ubfe(wr, off, cb);
			uint32_t ba = ubfe(wb, off, cb);
			if (e < 4) ba |= ubfe(wa, offa, ab) << 16;		// modes with alpha have at most two subsets; ab = 0 reads 0
			const uint32_t pm = (uint32_t)sbfe(pw, pidx[e], 1u);			// 0 / ~0: this endpoint's P-bit
			x_rg[k] = or3(rg << g_c.y, pk_lshr_v(g_c.z, rg), pm & g_c.w);
			x_ba[k] = or3(pk_lshl_v(g_ba.x, ba), pk_lshr_v(g_ba.y, ba), and_or(pm, g_ba.z, g_ba.w));
			off = FIXED >= 0 ? off + cb : opaque(off + cb);		// running sums as plain adds (opaque: not re-derived as e * cb with shifts)
			offa = FIXED >= 0 ? offa + ab : opaque(offa + ab);
		}
		uint4 row;
		row.x = (x_rg[0] << 8) | 0x00800080u;
		row.y = (x_ba[0] << 8) | 0x00800080u;
		row.z = pk_lshl_v(0x00020002u, pk_sub_u16(x_rg[1], x_rg[0]));	// 4*(e1-e0) per 16-bit lane (mod 2^16)
		row.w = pk_lshl_v(0x00020002u, pk_sub_u16(x_ba[1], x_ba[0]));
		lane.put_subset(s, row);
	}

	stage_priority<Tune::kBc7Prio, 1>();
	// subset number of texel i at bits 12-13 of (p_lo >> 2i) for i < 6, of (pword >> (2i - 12)) above: right shifts only
	const uint32_t pword = pe.pword, p_lo = pword << 12;

	// Colour index stream: 64 stream bits from its start; texels 0-7 consume `half` of them, texels 8-15 start
	// there.  Each window then gets the anchors' absent top bits inserted as zeros (w + (w & himask) doubles the
	// part of w at and above the insertion point), after which texel k of a window sits at bit k*ib.
	Group g_ci = L.template group<F_ROW_C / 4>();		// row_c, pos_c, ibc, imask_c
	Group g_cw = L.template group<F_WMUL_C / 4>();		// wmul_c, wadd_c, sel_comb (F_SEL_COMB), half_a
	L.pin(g_ci, g_cw);
	uint32_t c0, c1;
	lane.field64(g_ci.x, g_ci.y, c0, c1);
	const uint32_t route = pe.route;			// shifts use the low 5 bits of their amount
	uint32_t cw = c0, cw_hi = __builtin_amdgcn_alignbit(c1, c0, route >> 20);
	cw += cw & g_pj.w;
	const uint32_t ones = g_pj.z;
	cw += cw & (ones << (route & 31u));
	cw += cw & (ones << ((route >> 5) & 31u));
	cw_hi += cw_hi & (ones << ((route >> 10) & 31u));
	cw_hi += cw_hi & (ones << ((route >> 15) & 31u));
	const uint32_t ibc = g_ci.z, imask_c = g_ci.w, wmul_c = g_cw.x, wadd_c = g_cw.y;

	constexpr int kGroup = FIXED >= 0 ? Tune::kBc7UniformTexelGroup : 1;
	// one wave-uniform branch around two straight-line loops (a per-texel branch costs more than it skips)
	if (any_two) {
		// alpha stream (modes 4/5: one subset, the only anchor is texel 0)
		Group g_ai = L.template group<F_ROW_A2 / 4>();	// row_a2, pos_a2, ib4_c, ib4_a
		Group g_aw = L.template group<F_APAIR / 4>();	// apair, imask_pair, himask0_a, ib_pair
		L.pin(g_ai, g_aw);
		uint32_t a0, a1;
		lane.field64(g_ai.x, g_ai.y, a0, a1);
		uint32_t aw = a0, aw_hi = __builtin_amdgcn_alignbit(a1, a0, g_cw.w);
		aw += aw & g_aw.z;
		// Both streams in lockstep (round 4): quarter window q = {colour indices, alpha indices} of texels 4q .. 4q+3 in the 16-bit lanes
		// of ONE register.  Per texel: one `and` isolates both indices, one v_pk_mad_u16 turns them into both weights scaled by 256
		// (weight = (index * m + 128) >> 8 with m = 5461 / 2341 / 1092 for 2 / 3 / 4-bit indices: exact, bptc_weight16_mul), one packed
		// shift drops the scale, one packed shift advances both streams by their own widths -- four instructions for what two separate
		// streams took seven (and, multiply-add, shift, twice, and a v_perm pairing the weights): 41 -> 34 priced cycles per texel.
		const uint32_t sel_comb = g_cw.z, apair = g_aw.x, mpair = g_aw.y, ibpair = g_aw.w;
		uint32_t round128 = 0x00800080u;
		pin_vgpr(round128);				// (VOP3P takes no literal: a VGPR, not an SGPR source at half rate)
		uint32_t comb[4];
		comb[0] = perm(aw, cw, sel_comb);
		comb[1] = perm(aw >> g_ai.w, cw >> g_ai.z, sel_comb);
		comb[2] = perm(aw_hi, cw_hi, sel_comb);
		comb[3] = perm(aw_hi >> g_ai.w, cw_hi >> g_ai.z, sel_comb);
		// texels in groups: the group's subset rows are requested together, so the wave waits for LDS once per group.  (Groups of
		// 1 / 2 / 4 in the mixed-mode path measured 54.5 / 54.7 / 56.7 us on stream U in round 3: it waits per texel, its registers
		// are scarce; the uniform-wave copies gain a little from four in flight.)
#pragma unroll
		for (int i0 = 0; i0 < 16; i0 += kGroup) {
			uint4 s[kGroup];
#pragma unroll
			for (int j = 0; j < kGroup; j++) { const int i = i0 + j; s[j] = lane.get_subset(i < 6 ? p_lo >> (2 * i) : pword >> (2 * i - 12)); }
#pragma unroll
			for (int j = 0; j < kGroup; j++) {
				const int i = i0 + j;
				if (i == 8) stage_priority<Tune::kBc7Prio, 2>();
				uint32_t &q = comb[i >> 2];
				const uint32_t w = pk_lshr16(pk_mad_u16(q & mpair, apair, round128), 8);	// {colour weight, alpha weight}
				if ((i & 3) != 3) q = pk_lshr_v(ibpair, q);
				d[i] = perm(pk_mad_u16(s[j].w, w, s[j].y), pk_mad_u16_blo(s[j].z, w, s[j].x), gather);
			}
		}
	} else {
#pragma unroll
		for (int i0 = 0; i0 < 16; i0 += kGroup) {
			uint4 s[kGroup];
#pragma unroll
			for (int j = 0; j < kGroup; j++) { const int i = i0 + j; s[j] = lane.get_subset(i < 6 ? p_lo >> (2 * i) : pword >> (2 * i - 12)); }
#pragma unroll
			for (int j = 0; j < kGroup; j++) {
				const int i = i0 + j;
				if (i == 8) { cw = cw_hi; stage_priority<Tune::kBc7Prio, 2>(); }
				const uint32_t tc = DETEX_UMUL24(cw & imask_c, wmul_c) + wadd_c;
				cw >>= ibc;
				d[i] = perm(pk_mad_u16_bhi(s[j].w, tc, s[j].y), pk_mad_u16_bhi(s[j].z, tc, s[j].x), gather);
			}
		}
	}
}

// record of a block: its mode, +4 for mode 4 with the index-selection bit (block bit 7) set
DH uint32_t bc7_record_index(uint32_t first_dword) {
	const uint32_t mode = (uint32_t)__builtin_ctz(first_dword | 0x100u);
	return (first_dword & 0x9Fu) == 0x90u ? 8u : mode;		// ...1 0000 with bit 7 set
}

// UNIFORM: waves whose blocks all share one record run a copy specialised for it (kernels that are not on the
// throughput path -- clipped geometry, the checked per-block batch -- instantiate the plain form to bound code size)
template <bool UNIFORM> struct DecBPTCT {
	static constexpr int kBlockBytes = 16, kPixelBytes = 4;
	// <= 80 VGPRs: six waves per SIMD.  Same-run measurements on stream U / C: 64 VGPRs (9 dwords spilled) 65 / 52.4 us;
	// 72 VGPRs 59.1 / 52.7 before the next tile's block was really prefetched, 59.1 / 52.2 with it (two dwords spilled,
	// +2.5 % HBM traffic); 80 VGPRs (no spill) 57.9 / 50.8.  LDS (<= 20 KiB per workgroup) would admit eight workgroups.
	static constexpr int kWavesPerSimd = Tune::kBc7WavesPerSimd;
	// One workgroup per tile, like every other decoder.  A persistent grid (workgroups looping over tiles, the tables copied
	// once per resident workgroup, the next tile's block prefetched) was the better choice while the tables were 7.5 KiB
	// (58.8 vs 62.0 us, stream U); with today's 3.6 KiB it is the worse one -- same run, 8192^2, streams U / C: persistent
	// 57.4-57.9 / 50.7-50.8 us, one tile per workgroup 55.0 / 48.8 (tools/ab/kernels_persistent.h keeps that kernel for the
	// measurement build).  The block-major kernel likewise (U / M / C: 62.5-63.3 / 63.8-64.7 / 52.0 vs 64.5 / 66.1 / 47.4).
	static DH void prepare() { bc7_prepare(); }
	// the block-major exchange (kernels.h: decode_blocks) stages inside this decoder's own, by then dead, lane rows
	static constexpr bool kOwnStage = Tune::kBc7OwnStage;
	static DH void *stage_slot(uint32_t k, uint32_t b) { return bc7_stage_slot(k, b); }
	static DH uint32_t stage_step(uint32_t k) { return bc7_stage_step(k); }
	// A block that fails -- reserved mode (decompress-bptc.c:229-237, 361) or, in the checked form, a mode outside
	// mode_mask / the opaque flags (:363-369) -- is replaced by the mode-6 block whose other bits are all 0: endpoints,
	// P-bits and indices 0, which decodes to sixteen zero pixels on the normal path.  A lane-divergent early return made
	// every wave initialise sixteen result registers to zero first (and nearly every wave of a random stream holds a
	// reserved block or none at all -- the moves ran either way).
	static constexpr bool kZeroOnFailure = true;
	template <bool CHECKED> static DH bool decode(uint4 blk, uint32_t mode_mask, uint32_t flags, uint32_t (&d)[16]) {
		bool valid = (blk.x & 0xFFu) != 0u;
		if (CHECKED) {
			const uint32_t mode = (uint32_t)__builtin_ctz(blk.x | 0x100u);
			valid = valid && (mode_mask & (1u << mode)) != 0u && !(mode >= 4u && (flags & kFlagOpaqueOnly)) &&
				!(mode < 4u && (flags & kFlagNonOpaqueOnly));
		}
		const uint32_t keep = cond_to_mask(valid);
		blk.x = bfi(keep, blk.x, 0x40u); blk.y &= keep; blk.z &= keep; blk.w &= keep;
		const uint32_t r = bc7_record_index(blk.x);
		if (UNIFORM && !CHECKED) {
			const uint32_t r0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
			if (__builtin_amdgcn_ballot_w64(r != r0) == 0) {
				switch (r0) {
				case 0: bc7_decode_with<0>(blk, r, flags, d); return valid;
				case 1: bc7_decode_with<1>(blk, r, flags, d); return valid;
				case 2: bc7_decode_with<2>(blk, r, flags, d); return valid;
				case 3: bc7_decode_with<3>(blk, r, flags, d); return valid;
				case 4: bc7_decode_with<4>(blk, r, flags, d); return valid;
				case 5: bc7_decode_with<5>(blk, r, flags, d); return valid;
				case 6: bc7_decode_with<6>(blk, r, flags, d); return valid;
				case 7: bc7_decode_with<7>(blk, r, flags, d); return valid;
				default: bc7_decode_with<8>(blk, r, flags, d); return valid;
				}
			}
		}
		bc7_decode_with<-1>(blk, r, flags, d);
		return valid;
	}
};
using DecBPTC = DecBPTCT<Tune::kBc7Uniform>;
using DecBPTCPlain = DecBPTCT<false>;

}  // namespace detexhip
