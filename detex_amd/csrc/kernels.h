// kernels.h -- the gfx950 launch skeletons shared by every block format.
//
// Mapping: ONE LANE = ONE 4x4 BLOCK, a wavefront covers 64 consecutive
// blocks of the row-major block stream, a 256-thread workgroup 256 of them.
//   load   lane i reads block i: 64 lanes x 8/16 B = one 512 B / 1 KiB fully coalesced
//          global_load_dwordx2/x4 per wave, no LDS needed because nothing is shared
//   decode 16 texels in registers (4*P dwords, P = bytes per native pixel)
//   epilogue (optional, section 8f-2): in-register pixel-format conversion of the decoded block
//          (R<->B swizzle, RGBX8 -> packed RGB8) -- free at HBM rate, no second pass over the image
//   store  linear layout: texel row r of the 64 blocks is 64 x 4T contiguous bytes of image
//          row 4*by+r (T = bytes per target pixel)  ->  four (T=8: eight) wave-wide
//          non-temporal stores (cache policy `sc1 nt`, or `nt` where that measured better: store_with_policy),
//          each a single 0.75-2 KiB contiguous run (full 128 B lines, no partial-line writes, no read-for-ownership)
//          tiled layout: a lane's block is 16*T contiguous bytes and the wave's 64 blocks are
//          contiguous too, so they are staged in LDS in output order and leave as T/1 wave-wide
//          1 KiB-run stores as well (decode_blocks)
//   tables decoders with format tables copy them from __constant__ memory into LDS once per
//          workgroup (prepare_tables<Dec>() at kernel entry, dev_common.h)
// Invalid blocks are zero-filled and raise *status (texture.c:125-128 semantics): one relaxed
// agent-scope load + (only while it still reads 0) one store per wave, never an atomic RMW.
#pragma once
#include <type_traits>
#include "dev_common.h"
#include "path_types.h"

namespace detexhip {

// waves per SIMD the register allocation must leave room for (Dec::kWavesPerSimd; default: no constraint)
template <class Dec, class = void> struct WavesPerSimd { static constexpr int value = 1; };
template <class Dec> struct WavesPerSimd<Dec, std::enable_if_t<(Dec::kWavesPerSimd > 0)>> { static constexpr int value = Dec::kWavesPerSimd; };

// the same for kernels that add a staging array of their own to the decoder's LDS: four workgroups per CU are then all that fit,
// and asking the register allocator for more waves than that only takes registers away
template <class Dec> struct WavesPerSimdStaged { static constexpr int value = WavesPerSimd<Dec>::value > 4 ? 4 : WavesPerSimd<Dec>::value; };

// decoders that deliver sixteen zero pixels themselves when they return false (Dec::kZeroOnFailure)
template <class Dec, class = void> struct ZeroOnFailure { static constexpr bool value = false; };
template <class Dec> struct ZeroOnFailure<Dec, std::enable_if_t<Dec::kZeroOnFailure>> { static constexpr bool value = true; };

// decoders whose per-lane LDS rows double as the staging area of the block-major exchange (Dec::kOwnStage, Dec::stage_slot)
template <class Dec, class = void> struct OwnStage { static constexpr bool value = false; };
template <class Dec> struct OwnStage<Dec, std::enable_if_t<Dec::kOwnStage>> { static constexpr bool value = true; };

// decoders that can hand over texel rows as they complete (Dec::kRowWise, Dec::decode_rows)
template <class Dec, class = void> struct RowWise { static constexpr bool value = false; };
template <class Dec> struct RowWise<Dec, std::enable_if_t<Dec::kRowWise>> { static constexpr bool value = true; };

// decoders whose linear kernel requests the block before the format tables are copied (Dec::kLoadBeforeTables; default: after the copy's barrier)
template <class Dec, class = void> struct LoadBeforeTables { static constexpr bool value = false; };
template <class Dec> struct LoadBeforeTables<Dec, std::enable_if_t<Dec::kLoadBeforeTables>> { static constexpr bool value = true; };

// blocks per lane in the linear fast path (Dec::kLaneBlocks; default 1): see decode_linear_grouped
template <class Dec, class = void> struct LaneBlocks { static constexpr int value = 1; };
template <class Dec> struct LaneBlocks<Dec, std::enable_if_t<(Dec::kLaneBlocks > 1)>> { static constexpr int value = Dec::kLaneBlocks; };

template <int BYTES> struct BlockWord;
template <> struct BlockWord<8> { using type = uint2; };
template <> struct BlockWord<16> { using type = uint4; };

// ---- in-register pixel-format epilogues (reference: detexConvertPixels, convert.c) ----------------------------
// What the reference's callers ask detexDecompressTextureLinear for (BGRA8 / BGRX8: validate.c:204-209, detex-view.c:182;
// RGB8: detex-convert.c:283-284) is produced inside the decode kernel, with the exact result of the conversion path
// detexMatchConversion picks (convert.c:885-1063):
//   RGBA8 / RGBX8 natives   R<->B swap keeps byte 3 (:37-52); RGB8 drops it (:671-684)
//   R8, RG8                 -> RGBX8 (R, G or 0, 0, 0xFF) (:219-243), then as above
//   R16, RG16               -> R8 / RG8 by (x + 127) * 255 / 65535 (:258-281), then as R8 / RG8
//   SIGNED_R16 / RG16       -> R16 / RG16 by + 32768 (:158-181), then as R16 / RG16 (no path to BGRA8 in the reference: refused)
//   FLOAT_RGBX16 (BC6H)     -> RGBX16 by lrintf(clamp01(f) * 65535 + 0.5) rounding down (half-float.c:304-312) -> RGBX8 by
//                              the same 16 -> 8 map (:299-313): one 64 KiB lookup table per device (kHalfToU8)
//                              FLOAT_BGRX16: swap halves 0 and 2 (:54-70)
// (the epilogue numbers kEpi... are in path_types.h: the host side of the library names them too)
template <class Dec, class = void> struct NativeOf { static constexpr int value = kNatRGBA8; };
template <class Dec> struct NativeOf<Dec, std::void_t<decltype(Dec::kNative)>> { static constexpr int value = Dec::kNative; };

// half bit pattern -> 8-bit component of the FLOAT_RGBX16 -> RGBX16 -> RGBX8 path; filled per device by the host side
// of the library before the first launch that needs it (formats_bptc_float.hip: ensure_half_table)
// (static: one copy per translation unit that uses it -- only the BC6H one does, and that is the copy its ensure_half_table fills)
static __device__ __attribute__((aligned(16))) uint8_t kHalfToU8[65536];
// The kernels look halves up in a WORKGROUP COPY of the table's live part: unsigned BC6H decodes to halves 0 .. 0x7BFF
// (never negative, never Inf / NaN: decompress-bptc-float.c:613-621), and every half from 1.0 = 0x3C00 up converts to 255,
// so entries 0 .. 0x3C00 with the index clamped cover all of it in 15 KiB of LDS.  48 lookups per block as per-lane
// gathers from the 64 KiB table in global memory made a wave-wide load instruction touch up to 64 cache lines (8192^2 BC6H
// -> BGRX8: 244 us); LDS serves the same gather from 32 banks.
constexpr uint32_t kHalfOne = 0x3C00u;
constexpr uint32_t kHalfLutBytes = (kHalfOne + 1u + 15u) & ~15u;
DH uint8_t *half_lut() { __shared__ __attribute__((aligned(16))) uint8_t table[kHalfLutBytes]; return table; }
DH void half_lut_prepare() {
	typedef uint32_t v4 __attribute__((ext_vector_type(4)));
	const v4 *src = reinterpret_cast<const v4 *>(kHalfToU8);
	v4 *dst = reinterpret_cast<v4 *>(half_lut());
	for (uint32_t k = threadIdx.x; k < kHalfLutBytes / 16u; k += 256u) dst[k] = src[k];
	__syncthreads();
}
DH uint32_t half_to_u8(uint32_t half_bits) { return half_lut()[half_bits < kHalfOne ? half_bits : kHalfOne]; }

// pixel i (0..15) of a decoded block as R | G << 8 | B << 16 | 0xFF << 24
template <int NC> DH uint32_t pixel_as_rgbx8(const uint32_t *d, int i) {
	if constexpr (NC == kNatR8) return perm(0u, d[i >> 2], 0x0D0C0C00u + (uint32_t)(i & 3));
	else if constexpr (NC == kNatRG8) return perm(0u, d[i >> 1], 0x0D0C0000u | ((uint32_t)(2 * (i & 1) + 1) << 8) | (uint32_t)(2 * (i & 1)));
	else if constexpr (NC == kNatR16 || NC == kNatSignedR16) {
		const uint32_t x = ((d[i >> 1] >> (16 * (i & 1))) & 0xFFFFu) ^ (NC == kNatSignedR16 ? 0x8000u : 0u);
		return component16_to_8(x) | 0xFF000000u;
	} else if constexpr (NC == kNatRG16 || NC == kNatSignedRG16) {
		const uint32_t w = d[i] ^ (NC == kNatSignedRG16 ? 0x80008000u : 0u);
		return component16_to_8(w & 0xFFFFu) | (component16_to_8(w >> 16) << 8) | 0xFF000000u;
	} else {	// kNatFloatRGBX16: pixel = {R | G << 16, B | X << 16}
		const uint32_t w0 = d[2 * i], w1 = d[2 * i + 1];
		return half_to_u8(w0 & 0xFFFFu) | (half_to_u8(w0 >> 16) << 8) | (half_to_u8(w1 & 0xFFFFu) << 16) | 0xFF000000u;
	}
}

template <int EPI, int P, int NC = kNatRGBA8> struct Epilogue;
template <int P, int NC> struct Epilogue<kEpiNone, P, NC> {
	static constexpr int kRowDwords = P;		// dwords per 4-pixel row of the target
	static DH void apply(const uint32_t (&d)[4 * P], uint32_t (&o)[4 * P]) {
#pragma unroll
		for (int k = 0; k < 4 * P; k++) o[k] = d[k];
	}
};
template <> struct Epilogue<kEpiSwapRB8, 4, kNatRGBA8> {
	static constexpr int kRowDwords = 4;
	static DH void apply(const uint32_t (&d)[16], uint32_t (&o)[16]) {
#pragma unroll
		for (int k = 0; k < 16; k++) o[k] = perm(d[k], d[k], 0x03000102u);
	}
};
// four RGBX8 pixels -> three dwords of packed RGB8
DH void pack_rgb8_row(const uint32_t *px, uint32_t *o) {
	o[0] = perm(px[1], px[0], 0x04020100u);	// R0 G0 B0 R1
	o[1] = perm(px[2], px[1], 0x05040201u);	// G1 B1 R2 G2
	o[2] = perm(px[3], px[2], 0x06050402u);	// B2 R3 G3 B3
}
template <> struct Epilogue<kEpiPackRGB8, 4, kNatRGBA8> {
	static constexpr int kRowDwords = 3;
	static DH void apply(const uint32_t (&d)[16], uint32_t (&o)[12]) {
#pragma unroll
		for (int r = 0; r < 4; r++) pack_rgb8_row(d + 4 * r, o + 3 * r);
	}
};
template <> struct Epilogue<kEpiSwapRB16, 8, kNatFloatRGBX16> {
	static constexpr int kRowDwords = 8;
	static DH void apply(const uint32_t (&d)[32], uint32_t (&o)[32]) {
#pragma unroll
		for (int k = 0; k < 16; k++) {			// pixel = {R|G<<16, B|X<<16}
			o[2 * k] = perm(d[2 * k + 1], d[2 * k], 0x03020504u);	// B, G
			o[2 * k + 1] = perm(d[2 * k + 1], d[2 * k], 0x07060100u);	// R, X
		}
	}
};
template <int P, int NC> struct Epilogue<kEpiToRGBX8, P, NC> {
	static constexpr int kRowDwords = 4;
	static DH void apply(const uint32_t (&d)[4 * P], uint32_t (&o)[16]) {
#pragma unroll
		for (int k = 0; k < 16; k++) o[k] = pixel_as_rgbx8<NC>(d, k);
	}
};
template <int P, int NC> struct Epilogue<kEpiToBGRX8, P, NC> {
	static constexpr int kRowDwords = 4;
	static DH void apply(const uint32_t (&d)[4 * P], uint32_t (&o)[16]) {
#pragma unroll
		for (int k = 0; k < 16; k++) { const uint32_t v = pixel_as_rgbx8<NC>(d, k); o[k] = perm(v, v, 0x03000102u); }
	}
};
template <int P, int NC> struct Epilogue<kEpiToRGB8, P, NC> {
	static constexpr int kRowDwords = 3;
	static DH void apply(const uint32_t (&d)[4 * P], uint32_t (&o)[12]) {
#pragma unroll
		for (int r = 0; r < 4; r++) {
			uint32_t px[4];
#pragma unroll
			for (int x = 0; x < 4; x++) px[x] = pixel_as_rgbx8<NC>(d, 4 * r + x);
			pack_rgb8_row(px, o + 3 * r);
		}
	}
};
template <class Dec, int EPI> using EpilogueOf = Epilogue<EPI, Dec::kPixelBytes, NativeOf<Dec>::value>;
// tables an epilogue needs in LDS: called by every kernel with all 256 threads right after prepare_tables<Dec>()
template <class Dec, int EPI> DH void prepare_epilogue() {
	if constexpr (NativeOf<Dec>::value == kNatFloatRGBX16 && EPI >= kEpiToRGBX8) half_lut_prepare();
}

// (store_with_policy<POLICY, DWORDS>(v, p): a streaming store with an explicit cache policy -- gfx950_prims.h)
// cache policy of a decoder's stores: DEFAULT (tune.h) unless that is `sc1 nt` and the decoder is one of those the sweep found better
// off with plain `nt` (Dec::kStorePolicy)
template <class Dec, int DEFAULT, class = void> struct PolicyFor { static constexpr int value = DEFAULT; };
template <class Dec, int DEFAULT> struct PolicyFor<Dec, DEFAULT, std::enable_if_t<(Dec::kStorePolicy >= 0)>> { static constexpr int value = DEFAULT == 6 ? Dec::kStorePolicy : DEFAULT; };
template <class Dec> using StorePolicy = PolicyFor<Dec, Tune::kStorePolicy>;
// one 4-pixel row of ROW dwords with cache policy POLICY (0 = ordinary stores)
template <int ROW, int POLICY> DH void store_row(uint8_t *dst, const uint32_t *d) {
	if constexpr (ROW == 1) {
		store_with_policy<POLICY, 1>(d[0], reinterpret_cast<uint32_t *>(dst));
	} else if constexpr (ROW == 2) {
		typedef uint32_t v2 __attribute__((ext_vector_type(2)));
		store_with_policy<POLICY, 2>(v2{ d[0], d[1] }, reinterpret_cast<v2 *>(dst));
	} else if constexpr (ROW == 3) {
		typedef uint32_t v3 __attribute__((ext_vector_type(3)));
		typedef v3 v3_unaligned __attribute__((aligned(4)));
		// (only dword-aligned; the alignment of the pointee type matters to the plain and builtin forms, not to the instruction)
		if constexpr (POLICY == 4) __builtin_nontemporal_store(v3{ d[0], d[1], d[2] }, reinterpret_cast<v3_unaligned *>(dst));
		else if constexpr (POLICY == 0) *reinterpret_cast<v3_unaligned *>(dst) = v3{ d[0], d[1], d[2] };
		else store_with_policy<POLICY, 3>(v3{ d[0], d[1], d[2] }, reinterpret_cast<v3 *>(dst));
	} else {
		typedef uint32_t v4 __attribute__((ext_vector_type(4)));
#pragma unroll
		for (int k = 0; k < ROW / 4; k++)
			store_with_policy<POLICY, 4>(v4{ d[4 * k], d[4 * k + 1], d[4 * k + 2], d[4 * k + 3] }, reinterpret_cast<v4 *>(dst) + k);
	}
}

// one texel row of ROW dwords to a destination that is only dword-aligned
template <int ROW> DH void store_row_dword_aligned(uint8_t *dst, const uint32_t *d) {
	typedef uint32_t v2 __attribute__((ext_vector_type(2)));
	typedef uint32_t v3 __attribute__((ext_vector_type(3)));
	typedef uint32_t v4 __attribute__((ext_vector_type(4)));
	typedef v2 v2a __attribute__((aligned(4)));
	typedef v3 v3a __attribute__((aligned(4)));
	typedef v4 v4a __attribute__((aligned(4)));
	if constexpr (ROW == 1) __builtin_nontemporal_store(d[0], reinterpret_cast<uint32_t *>(dst));
	else if constexpr (ROW == 2) __builtin_nontemporal_store(v2{ d[0], d[1] }, reinterpret_cast<v2a *>(dst));
	else if constexpr (ROW == 3) __builtin_nontemporal_store(v3{ d[0], d[1], d[2] }, reinterpret_cast<v3a *>(dst));
	else {
#pragma unroll
		for (int k = 0; k < ROW / 4; k++)
			__builtin_nontemporal_store(v4{ d[4 * k], d[4 * k + 1], d[4 * k + 2], d[4 * k + 3] }, reinterpret_cast<v4a *>(dst) + k);
	}
}

// measurement builds only: a pause between a wave's row stores (Tune::kStoreSleep, 0 in the product)
DH void store_pause() { sleep_cycles64<Tune::kStoreSleep>(); }
// raise the "some block was invalid" word without an atomic RMW storm (see header comment)
DH void raise_status(bool bad, uint32_t *status) {
	if (status == nullptr) return;
	if (__builtin_amdgcn_ballot_w64(bad) == 0) return;		// wave-uniform
	if (bad && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
		__hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// block index -> (block row, block column).  Texture widths are almost always powers of two, where
// this is a shift and a mask on a wave-uniform (kernel-argument) width; the generic 32-bit division
// (a ~20-instruction v_mul_hi_u32 sequence) is kept off the common path.
DH void split_index(uint32_t i, uint32_t width_in_blocks, uint32_t &by, uint32_t &bx) {
	if ((width_in_blocks & (width_in_blocks - 1u)) == 0u) {
		by = i >> __builtin_ctz(width_in_blocks);
		bx = i & (width_in_blocks - 1u);
	} else {
		by = i / width_in_blocks;
		bx = i - by * width_in_blocks;
	}
}

// 64-bit pixels: a lane's texel row is 32 B, so a plain dwordx4 store writes 16 of every 32 bytes and the
// line is completed by the NEXT instruction -- measured 2.5 TB/s for streaming stores (BC6H 217 us).  Each row
// is transposed through LDS (2.1 KiB per wave) so that every store instruction covers contiguous 1 KiB runs,
// as for the 32-bit formats.  Works for any width: output vector e of the wave belongs to block first + e/2,
// whose place in the image is recomputed by the storing lane (a wave that straddles the end of a block row
// simply continues on the next one); every lane of the wave must call this, `live` or not.
struct WideRowStore {
	typedef uint32_t v4 __attribute__((ext_vector_type(4)));
	// [half of the lane's 32-byte row][lane], the second half 72 vectors on: writes (consecutive lanes) and
	// transposed reads (output vector e = 2*lane' + half) are both bank-conflict free
	static constexpr int STRIDE = 72;
	v4 *slab;
	uint32_t lane, src_a, src_b;
	uint8_t *dst_a, *dst_b;
	uint64_t pitch;
	bool store_a, store_b, full;
	DH WideRowStore(uint8_t *pixels, uint64_t pitch_, uint32_t width_in_blocks, uint32_t first, uint32_t n_blocks) : pitch(pitch_) {
		__shared__ v4 xpose[4][STRIDE + 64];
		slab = xpose[threadIdx.x >> 6];
		lane = threadIdx.x & 63u;
		src_a = (lane & 1u) * STRIDE + (lane >> 1); src_b = src_a + 32u;	// vectors e = lane and e = 64 + lane
		// destinations of those two vectors: blocks first + lane/2 and first + 32 + lane/2
		const uint32_t ia = first + (lane >> 1), ib = ia + 32u;
		uint32_t by, bx;
		// power-of-two widths of at least 64 blocks (kernel-argument uniform: a scalar branch): the wave lies in ONE block row, the
		// second vector's place is the first one's + 1 KiB -- no second index split, no second 64-bit multiply-add (and the general
		// path's reciprocal stays out of this one)
		if ((width_in_blocks & (width_in_blocks - 1u)) == 0u && width_in_blocks >= 64u) {
			by = ia >> __builtin_ctz(width_in_blocks); bx = ia & (width_in_blocks - 1u);
			dst_a = pixels + (uint64_t)(by * 4u) * pitch + (uint64_t)bx * 32u + (lane & 1u) * 16u;
			dst_b = dst_a + 1024;
		} else {
			split_index(ia, width_in_blocks, by, bx);
			dst_a = pixels + (uint64_t)(by * 4u) * pitch + (uint64_t)bx * 32u + (lane & 1u) * 16u;
			split_index(ib, width_in_blocks, by, bx);
			dst_b = pixels + (uint64_t)(by * 4u) * pitch + (uint64_t)bx * 32u + (lane & 1u) * 16u;
		}
		store_a = ia < n_blocks; store_b = ib < n_blocks;
		// every wave but the stream's last stores all of its 128 vectors: a scalar test, unguarded stores
		full = wave_full(first, n_blocks);
	}
	static DH bool wave_full(uint32_t first, uint32_t n_blocks) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)first) + 64u <= n_blocks; }
	// the next texel row of the wave through the transpose (rows must come in order 0..3); `mine`: this lane has a decoded row to
	// contribute.  Returns the two output vectors this lane stores for the row.
	DH void exchange(bool mine, const uint32_t *o, v4 &a, v4 &b) {
		if (mine) {
			slab[lane] = v4{ o[0], o[1], o[2], o[3] };
			slab[STRIDE + lane] = v4{ o[4], o[5], o[6], o[7] };
		}
		// same wave: LDS operations complete in order; the wavefront-scope fences only keep the
		// compiler from reordering or forwarding across the exchange (they emit no cache traffic)
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		a = slab[src_a]; b = slab[src_b];
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
	}
	DH void store(const v4 &a, const v4 &b) {
		if (full || store_a) store_with_policy<Tune::kStorePolicyWide, 4>(a, reinterpret_cast<v4 *>(dst_a));
		store_pause();
		if (full || store_b) store_with_policy<Tune::kStorePolicyWide, 4>(b, reinterpret_cast<v4 *>(dst_b));
		store_pause();
		dst_a += pitch; dst_b += pitch;			// (one 64-bit add each; r * pitch came out as two v_mad_u64_u32 per pointer)
	}
	DH void row(bool mine, const uint32_t *o) { v4 a, b; exchange(mine, o, a, b); store(a, b); }
};
DH void store_rows_wide_pixels(uint8_t *pixels, uint64_t pitch, uint32_t width_in_blocks, uint32_t first, uint32_t n_blocks,
		bool live, const uint32_t (&o)[32]) {
	WideRowStore st(pixels, pitch, width_in_blocks, first, n_blocks);
	if constexpr (Tune::kWideBurst) {
		// all four rows through the transpose first, then the wave's eight stores back to back: a row's stores used to wait for that
		// row's LDS round trip, which spread the eight stores over the four exchanges -- and the write path wants a wave's stores
		// in one burst (profiles/AB_RECORD.md: stores spread over a wave's life cost BC6H 13 %).  Register-neutral: a row's eight
		// result dwords die as its two output vectors are born.
		WideRowStore::v4 a[4], b[4];
#pragma unroll
		for (int r = 0; r < 4; r++) st.exchange(live, o + 8 * r, a[r], b[r]);
		// ... and the four waves of a workgroup start their bursts APART (wave w waits w * kWideStagger * 64 cycles): together they
		// write 32 KiB into the same four image rows, and on coherent content -- where all four finish decoding in the same cycle --
		// one after the other is faster: BC6H fixture 84.4-85.3 -> 81.5-82.4 us, random blocks unchanged (82.7-83.6 against 83.1-83.4);
		// the kernels with pixels up to 32 bits (four stores per wave) gain nothing from the same delay, nor does a stagger at the
		// start of the kernel (profiles/AB_RECORD.md)
		if constexpr (Tune::kWideStagger > 0) {
			const uint32_t w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
			for (uint32_t k = 0; k < w; k++) sleep_cycles64<Tune::kWideStagger>();
		}
#pragma unroll
		for (int r = 0; r < 4; r++) st.store(a[r], b[r]);
	} else {
#pragma unroll
		for (int r = 0; r < 4; r++) st.row(live, o + 8 * r);
	}
}

// The block of lane i as ONE 8/16-byte load: left to itself the compiler loads the first dword, tests the decoders'
// early-outs on it and only then fetches the rest -- two dependent HBM round trips per block.  pin_block() makes all
// dwords of the word live at one point (an empty asm; no instruction), which keeps the load whole; it is also where
// the wave waits for the data, so a caller can put work between the request and the pin.
template <class Word> DH void pin_block(Word &blk) {
	if constexpr (sizeof(Word) == 16) pin_vgpr(blk.x, blk.y, blk.z, blk.w);
	else pin_vgpr(blk.x, blk.y);
}
template <class Dec> DH typename BlockWord<Dec::kBlockBytes>::type load_block(const void *blocks, uint32_t i) {
	using Word = typename BlockWord<Dec::kBlockBytes>::type;
	Word blk = reinterpret_cast<const Word *>(blocks)[i];
	pin_block(blk);
	return blk;
}

// `decode_flags` of the texture-driver kernels: not the reference's flags (its drivers pass none, texture.c:88,123) but this
// library's spec-conformance switches for the two BPTC quirks (bptc_common.h: kFlagSpec...; detexhipSetQuirks), a wave-uniform
// kernel argument that every other decoder ignores.
// decode + zero-fill on failure + epilogue; returns ok
template <class Dec, int EPI, bool CHECKED>
DH bool decode_word(const typename BlockWord<Dec::kBlockBytes>::type &blk, uint32_t mode_mask, uint32_t flags,
		uint32_t (&o)[4 * EpilogueOf<Dec, EPI>::kRowDwords]) {
	constexpr int P = Dec::kPixelBytes;
	uint32_t d[4 * P];
	bool ok = true;
	if constexpr (Tune::kNoCompute) {	// measurement build: memory traffic without the decode
#pragma unroll
		for (int k = 0; k < 4 * P; k++) d[k] = (k & 1) ? blk.y : blk.x;
	} else {
		ok = Dec::template decode<CHECKED>(blk, mode_mask, flags, d);
	}
	EpilogueOf<Dec, EPI>::apply(d, o);
	// texture.c:125-128: a failed block is zero-filled in the TARGET format (not "converted zeros": X / alpha stay 0).
	// Decoders that already deliver zeros for a failed block (Dec::kZeroOnFailure) skip this when the epilogue maps zeros to
	// zeros (native target, channel swaps).
	constexpr bool already_zero = ZeroOnFailure<Dec>::value && (EPI == kEpiNone || EPI == kEpiSwapRB16 || EPI == kEpiSwapRB8);
	if (!already_zero && !ok) {
#pragma unroll
		for (int k = 0; k < 4 * EpilogueOf<Dec, EPI>::kRowDwords; k++) o[k] = 0u;
	}
	return ok;
}
template <class Dec, int EPI, bool CHECKED>
DH bool decode_block(const void *blocks, uint32_t i, uint32_t mode_mask, uint32_t flags,
		uint32_t (&o)[4 * EpilogueOf<Dec, EPI>::kRowDwords]) {
	return decode_word<Dec, EPI, CHECKED>(load_block<Dec>(blocks, i), mode_mask, flags, o);
}
// always true in the product build; the "decode without its stores" measurement build makes it practically never true
DH bool stores_enabled(const uint32_t *o) {
	if constexpr (Tune::kNoStore) return o[0] == 0x9E3779B9u && o[1] == 0x7F4A7C15u;
	else return true;
}

// ---- linear layout, fast path: width % 4 == 0, vector-aligned rows ----------------------------
// One workgroup per tile of 256 consecutive blocks, dispatched by the hardware.  (A persistent grid -- workgroups looping
// over tiles, tables copied once per resident workgroup, next tile's block prefetched -- measured 1-20 % slower for every
// format, BC7 included once its tables had shrunk: profiles/AB_RECORD.md; that kernel lives in tools/ab/kernels_persistent.h.)
template <class Dec, int EPI, bool NT>
__global__ __launch_bounds__(256, WavesPerSimd<Dec>::value) void decode_linear(const void *__restrict__ blocks,
		uint8_t *__restrict__ pixels, uint32_t width_in_blocks, uint32_t n_blocks, uint64_t pitch,
		uint32_t *__restrict__ status, uint32_t decode_flags) {
	constexpr int ROW = EpilogueOf<Dec, EPI>::kRowDwords;
	using Word = typename BlockWord<Dec::kBlockBytes>::type;
	constexpr bool kEarly = Tune::kLoadBeforeTables || LoadBeforeTables<Dec>::value;	// (the Tune switch: every decoder, measurement builds)
	Word early;
	if constexpr (kEarly) {
		const uint32_t i0 = blockIdx.x * 256u + threadIdx.x;
		early = reinterpret_cast<const Word *>(blocks)[i0 < n_blocks ? i0 : n_blocks - 1u];
	}
	prepare_tables<Dec>();
	prepare_epilogue<Dec, EPI>();
	// The block is loaded AFTER the table copy's barrier: requested before it, the barrier waits for the workgroup's slowest
	// HBM round trip (SIGNED_RGTC2 46.7 -> 50.3 us, EAC_R11 24.3 -> 25.7 in the same run) -- except for the decoders that say
	// otherwise (LoadBeforeTables: unsigned BC6H, whose long decode hides the wait and whose blocks out of HBM arrive 4 % sooner).  The load is unconditional, with
	// the index clamped into the stream: a load under a branch makes the compiler wait for it at the end of the branch.
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	Word blk;
	ReadAheadSink<Word> ahead;
	if constexpr (Tune::kPrefetchTiles > 0) {	// MEASUREMENT BUILDS ONLY (Tune::kPrefetchTiles): the blocks of a later tile requested too, and dropped
		const uint32_t j = i + (uint32_t)Tune::kPrefetchTiles * 256u;
		load_with_read_ahead(reinterpret_cast<const Word *>(blocks) + (i < n_blocks ? i : n_blocks - 1u), reinterpret_cast<const Word *>(blocks) + (j < n_blocks ? j : n_blocks - 1u), blk, ahead);
	} else if constexpr (kEarly) {
		blk = early;
	} else if constexpr (Tune::kLoadPolicy != 0) {	// MEASUREMENT BUILDS ONLY
		load_with_policy<Tune::kLoadPolicy>(reinterpret_cast<const Word *>(blocks) + (i < n_blocks ? i : n_blocks - 1u), blk);
	} else {
		blk = reinterpret_cast<const Word *>(blocks)[i < n_blocks ? i : n_blocks - 1u];
	}
	pin_block(blk);
	struct KeepAhead { const ReadAheadSink<Word> &w; DH ~KeepAhead() { if constexpr (Tune::kPrefetchTiles > 0) keep_alive(w); } } keep_ahead{ ahead };
	if constexpr (ROW == 8 && NT) {
		// 64-bit pixels: rows leave through the per-wave LDS transpose; lanes past the end stay for the exchange
		// (The branch around the decode costs ~30 v_mov: the compiler initialises the result registers for the lanes that skip it.
		// Decoding unconditionally -- the lanes past the end have a copy of the last block -- removes them, and with the branch
		// gone the scheduler treats decode and exchange as one region and allocates 126 VGPRs instead of 65: four workgroups per
		// CU instead of seven, BC6H on coherent content 80 -> 91 us.  The branch stays.)
		const bool live = i < n_blocks;
		if constexpr (RowWise<Dec>::value && EPI == kEpiNone && !Tune::kNoCompute && !Tune::kNoStore && Tune::kRowWise) {
			// MEASUREMENT BUILDS ONLY (Tune::kRowWise): each texel row is exchanged and stored as soon as the decoder has it,
			// while the following rows are still being interpolated -- the 32 result dwords are never all alive (28 VALU operations
			// fewer per wave), and the wave's eight store instructions are spread over the second half of its life.  That spreading
			// is the wrong thing to do: 8192^2 BC6H 93.6 / 93.7 / 94.1 us (U / M / C) against 82.8 / 83.3 / 86.2 for the eight stores
			// in one burst at the end (profiles/AB_RECORD.md).  Every lane decodes here (those past the end a copy of the last
			// block) because every lane stores: output vector e of the wave is written by lane e whoever decoded it.
			WideRowStore st(pixels, pitch, width_in_blocks, i - (threadIdx.x & 63u), n_blocks);
			const bool ok = Dec::template decode_rows<false>(blk, 0xFFFFFFFFu, decode_flags, [&](int, const uint32_t (&row)[8]) { st.row(true, row); });
			if (live) raise_status(!ok, status);
		} else {
			uint32_t o[4 * ROW];
			bool ok = true;
			if (live) ok = decode_word<Dec, EPI, false>(blk, 0xFFFFFFFFu, decode_flags, o);
			if (stores_enabled(o))
				store_rows_wide_pixels(pixels, pitch, width_in_blocks, i - (threadIdx.x & 63u), n_blocks, live, o);
			if (live) raise_status(!ok, status);
		}
	} else {
		if (i >= n_blocks) return;
		uint32_t o[4 * ROW];
		const bool ok = decode_word<Dec, EPI, false>(blk, 0xFFFFFFFFu, decode_flags, o);
		uint32_t by, bx;
		split_index(i, width_in_blocks, by, bx);
		uint8_t *dst = pixels + (uint64_t)(by * 4u) * pitch + (uint64_t)bx * (4u * ROW);
		if (stores_enabled(o)) {
#pragma unroll
			for (int r = 0; r < 4; r++) { store_row<ROW, (NT ? StorePolicy<Dec>::value : 0)>(dst + (uint64_t)r * pitch, o + r * ROW); store_pause(); }
		}
		raise_status(!ok, status);
	}
}

// ---- linear layout, any dword-aligned geometry: rows staged per workgroup, stores aligned to 64-byte sectors ---------------
// A streaming store instruction whose 1 KiB run does not start on a 64-byte boundary leaves a partial sector at both ends, to
// be completed by the neighbouring wave's store -- measured on this chip (tools/ubench/hbm_ref.hip, 256 MiB fill): runs at
// offsets 4 .. 48 and 112 from a 64-byte boundary 4.4-4.5 TB/s, at offset 0 5.8, at offset 64 5.4.  decode_linear's runs
// start on such boundaries only when every image row does (row bytes, pitch and base multiples of 64): a width like 8188 (row
// = 32752 bytes), and every width that is not a multiple of four (clipped last block column, texture.c:116-120, 132-136; rows
// 4 / 8 / 12 bytes off 16), lost 30-35 % (BC1 8188 x 8192 60.7 us, 8190 x 8190 63-64 us against 43 for 8192 x 8192).
// Here a workgroup decodes up to 256 blocks of ONE block row, parks the four texel rows in LDS, and then writes each
// texel row's byte range [x0, x1) of its image row in 16-byte vectors laid on the 64-byte grid of the ADDRESS SPACE (wave w
// covers the w-th KiB from the first sector boundary in the range), whatever the range's own alignment: only the first and
// the last sector of a 4 KiB piece can be partial, and each is one store instruction.  The image's right and bottom edges are clipped by the range, so clipped
// textures need no separate edge pass.  Needs rows that are dword-aligned and a whole number of dwords long; anything else
// (R8 / RG8 / RGB8 images of odd width) goes pixel by pixel through decode_linear_clipped.
template <class Dec, int EPI>
__global__ __launch_bounds__(256, WavesPerSimdStaged<Dec>::value) void decode_linear_staged(const void *__restrict__ blocks,
		uint8_t *__restrict__ pixels, uint32_t width_in_blocks, uint32_t row_bytes, uint32_t height, uint64_t pitch,
		uint32_t *__restrict__ status, uint32_t tiles_per_row, uint32_t decode_flags) {
	constexpr int ROW = EpilogueOf<Dec, EPI>::kRowDwords;
	constexpr uint32_t PIECE = 4u * ROW;				// bytes of one block in one texel row
	constexpr uint32_t SLACK = 8u, ROW_DWORDS = 256u * ROW + 2u * SLACK;	// a staged texel row: 32 bytes of slack in front and behind
	typedef uint32_t v4 __attribute__((ext_vector_type(4)));
	using Word = typename BlockWord<Dec::kBlockBytes>::type;
	__shared__ __attribute__((aligned(16))) uint32_t stage[4][ROW_DWORDS];
	prepare_tables<Dec>();
	prepare_epilogue<Dec, EPI>();
	const uint32_t by = blockIdx.x / tiles_per_row, tile = blockIdx.x - by * tiles_per_row;
	const uint32_t bx = tile * 256u + threadIdx.x;
	const bool live = bx < width_in_blocks;
	Word blk = reinterpret_cast<const Word *>(blocks)[by * width_in_blocks + (live ? bx : width_in_blocks - 1u)];
	pin_block(blk);
	uint32_t o[4 * ROW];
	bool ok = true;
	if (live) {
		ok = decode_word<Dec, EPI, false>(blk, 0xFFFFFFFFu, decode_flags, o);
#pragma unroll
		for (int r = 0; r < 4; r++) {
			uint32_t *slot = &stage[r][SLACK + threadIdx.x * ROW];
			if constexpr (ROW % 4 == 0) {
#pragma unroll
				for (int k = 0; k < ROW / 4; k++) reinterpret_cast<v4 *>(slot)[k] = v4{ o[r * ROW + 4 * k], o[r * ROW + 4 * k + 1], o[r * ROW + 4 * k + 2], o[r * ROW + 4 * k + 3] };
			} else {
#pragma unroll
				for (int k = 0; k < ROW; k++) slot[k] = o[r * ROW + k];
			}
		}
	}
	__syncthreads();
	if (live) raise_status(!ok, status);
	const uint32_t x0 = tile * 256u * PIECE;				// the tile's byte range in its image rows, clipped at the image
	const uint32_t grid_end = width_in_blocks * PIECE, x_end = row_bytes < grid_end ? row_bytes : grid_end;
	const uint32_t x1 = x0 + 256u * PIECE < x_end ? x0 + 256u * PIECE : x_end;
	if (x1 <= x0) return;
#pragma unroll
	for (int r = 0; r < 4; r++) {
		const uint32_t y = by * 4u + (uint32_t)r;
		if (y >= height) break;
		uint8_t *out = pixels + (uint64_t)y * pitch + x0;			// the tile's piece of image row y: n bytes, dword-aligned
		const uint32_t *staged = &stage[r][SLACK];
		const uint32_t n = x1 - x0;
		// [head: up to the first 64-byte boundary][body: whole sectors][tail].  Head and tail are each written by ONE
		// dword-store instruction (lane k its k-th dword), so a boundary sector receives one partial write from this workgroup
		// and one from its neighbour -- per-dword instructions made every one of them a partial write of its own.
		const uint32_t lead = (uint32_t)(reinterpret_cast<uintptr_t>(out) & 63u);
		const uint32_t head = lead ? (64u - lead < n ? 64u - lead : n) : 0u;
		const uint32_t body = (n - head) & ~63u, tail = n - head - body;
		if (threadIdx.x < head / 4u) __builtin_nontemporal_store(staged[threadIdx.x], reinterpret_cast<uint32_t *>(out) + threadIdx.x);
		if (threadIdx.x >= 64u && threadIdx.x - 64u < tail / 4u)
			__builtin_nontemporal_store(staged[(head + body) / 4u + threadIdx.x - 64u], reinterpret_cast<uint32_t *>(out + head + body) + (threadIdx.x - 64u));
		const uint32_t shift = head & 12u;					// the body's position in the staged row modulo 16
		for (uint32_t c = threadIdx.x; c < body / 16u; c += 256u) {
			const uint32_t src = head + 16u * c;				// byte offset into the staged row
			// the two aligned 16-byte LDS vectors that contain the chunk (the slack keeps the second inside the array)
			const v4 *q = reinterpret_cast<const v4 *>(reinterpret_cast<const uint8_t *>(staged) + (src - shift));
			const v4 a = q[0], b = q[1];
			v4 d;
			if (shift == 0u) d = a;
			else if (shift == 4u) d = v4{ a.y, a.z, a.w, b.x };
			else if (shift == 8u) d = v4{ a.z, a.w, b.x, b.y };
			else d = v4{ a.w, b.x, b.y, b.z };
			__builtin_nontemporal_store(d, reinterpret_cast<v4 *>(out + src));
		}
	}
}

// ---- linear layout, fast path for narrow pixels: G horizontally adjacent blocks per lane ----------------------------
// With one block per lane a texel row of R8 pixels is 4 bytes, so a wave's store instruction covers only a 256-byte run
// (RGTC1 8192^2: 0.81 of the HBM rate even with the decode removed).  Here a lane decodes G consecutive blocks of one block
// row and writes G rows' worth per store: 1 KiB runs for RGTC1 (G = 4), and G independent block loads in flight per lane.
// Measured against the one-block kernel in the same run (8192^2, streams U / C): RGTC1 17.8 / 15.1 -> 15.0-15.7 / 14.0 us;
// SIGNED_RGTC1 (R16, G = 2: half as many table copies too) 27.3 / 24.7 -> 25.5 / 24.1.  The other 2-byte formats do NOT
// gain from G = 2 -- RGTC2 29.5 / 28.1 -> 30.1-31.2 / 29.3, EAC_R11 24.4 / 23.1 -> 25.1 / 23.3, EAC_SIGNED_R11 25.7 / 24.2
// -> 25.1 / 24.7: 512-byte runs already stream well and half the waves hide less latency -- and keep one block per lane.
// Needs width_in_blocks % G == 0 (the G blocks then share a block row) and rows aligned for the wider store; the host
// falls back to decode_linear otherwise.
template <class Dec, int EPI, bool NT, int G>
__global__ __launch_bounds__(256) void decode_linear_grouped(const void *__restrict__ blocks, uint8_t *__restrict__ pixels,
		uint32_t width_in_blocks, uint32_t n_blocks, uint64_t pitch, uint32_t *__restrict__ status, uint32_t decode_flags) {
	constexpr int ROW = EpilogueOf<Dec, EPI>::kRowDwords;
	static_assert(ROW * G == 2 || ROW * G == 4, "a lane writes one 8- or 16-byte vector per texel row");
	using Word = typename BlockWord<Dec::kBlockBytes>::type;
	prepare_tables<Dec>();
	prepare_epilogue<Dec, EPI>();
	const uint32_t n_groups = n_blocks / G, group = blockIdx.x * 256u + threadIdx.x;
	const uint32_t first = (group < n_groups ? group : n_groups - 1u) * G;	// unconditional load, index clamped into the stream
	Word blk[G];
#pragma unroll
	for (int g = 0; g < G; g++) blk[g] = reinterpret_cast<const Word *>(blocks)[first + g];
#pragma unroll
	for (int g = 0; g < G; g++) pin_block(blk[g]);
	if (group >= n_groups) return;
	uint32_t o[G][4 * ROW];
	bool ok = true;
#pragma unroll
	for (int g = 0; g < G; g++) ok &= decode_word<Dec, EPI, false>(blk[g], 0xFFFFFFFFu, decode_flags, o[g]);
	uint32_t by, bx;
	split_index(first, width_in_blocks, by, bx);
	uint8_t *dst = pixels + (uint64_t)(by * 4u) * pitch + (uint64_t)bx * (4u * ROW);
	if (stores_enabled(o[0])) {
#pragma unroll
		for (int r = 0; r < 4; r++) {
			uint32_t row[ROW * G];
#pragma unroll
			for (int g = 0; g < G; g++)
#pragma unroll
				for (int k = 0; k < ROW; k++) row[g * ROW + k] = o[g][r * ROW + k];
			store_row<ROW * G, (NT ? StorePolicy<Dec>::value : 0)>(dst + (uint64_t)r * pitch, row);
		}
	}
	raise_status(!ok, status);
}

// ---- linear layout, clipped / unaligned path (texture.c:116-120,132-136) ----------------------
// Any width/height/pitch/pointer alignment down to the pixel size; texels outside the image
// are dropped.  Per-pixel stores: this path is for edge geometry, not for throughput.
template <int ROW> DH void store_pixel(uint8_t *dst, const uint32_t *row, int x) {
	if constexpr (ROW == 1) *dst = (uint8_t)(row[0] >> (8 * x));
	else if constexpr (ROW == 2) *reinterpret_cast<uint16_t *>(dst) = (uint16_t)(row[x >> 1] >> (16 * (x & 1)));
	else if constexpr (ROW == 3) {
#pragma unroll
		for (int k = 0; k < 3; k++) dst[k] = (uint8_t)(row[(3 * x + k) >> 2] >> (8 * ((3 * x + k) & 3)));
	} else if constexpr (ROW == 4) *reinterpret_cast<uint32_t *>(dst) = row[x];
	else { reinterpret_cast<uint32_t *>(dst)[0] = row[2 * x]; reinterpret_cast<uint32_t *>(dst)[1] = row[2 * x + 1]; }
}

template <class Dec, int EPI>
__global__ __launch_bounds__(256) void decode_linear_clipped(const void *__restrict__ blocks,
		uint8_t *__restrict__ pixels, uint32_t width_in_blocks, uint32_t n_blocks, uint32_t width,
		uint32_t height, uint64_t pitch, uint32_t *__restrict__ status, uint32_t decode_flags) {
	constexpr int ROW = EpilogueOf<Dec, EPI>::kRowDwords;
	prepare_tables<Dec>();
	prepare_epilogue<Dec, EPI>();
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n_blocks) return;
	uint32_t o[4 * ROW];
	const bool ok = decode_block<Dec, EPI, false>(blocks, i, 0xFFFFFFFFu, decode_flags, o);
	uint32_t by, bx;
	split_index(i, width_in_blocks, by, bx);
	// blocks that lie completely inside the image write whole texel rows when rows are dword-aligned (vector stores
	// that need no more than that); the others, or every block when rows are not even dword-aligned, go pixel by pixel
	const bool rows_dword_aligned = ((reinterpret_cast<uintptr_t>(pixels) | pitch) & 3u) == 0;
	if (rows_dword_aligned && bx * 4u + 4u <= width && by * 4u + 4u <= height) {
		uint8_t *dst = pixels + (uint64_t)(by * 4u) * pitch + (uint64_t)bx * (4u * ROW);
#pragma unroll
		for (int r = 0; r < 4; r++) store_row_dword_aligned<ROW>(dst + (uint64_t)r * pitch, o + r * ROW);
	} else {
#pragma unroll
		for (int r = 0; r < 4; r++) {
			const uint32_t y = by * 4u + r;
			if (y >= height) continue;
			uint8_t *dst = pixels + (uint64_t)y * pitch + (uint64_t)bx * (4u * ROW);
#pragma unroll
			for (int x = 0; x < 4; x++)
				if (bx * 4u + x < width) store_pixel<ROW>(dst + x * ROW, o + r * ROW, x);
		}
	}
	raise_status(!ok, status);
}

// ---- block-major output (detexDecompressTextureTiled, texture.c:77-98) and the batched form of
// the per-block API (mode_mask / flags honoured, per-block ok byte) ----------------------------
template <class Dec, int EPI, bool CHECKED>
__global__ __launch_bounds__(256, (OwnStage<Dec>::value && EpilogueOf<Dec, EPI>::kRowDwords == 4 ? WavesPerSimd<Dec>::value : WavesPerSimdStaged<Dec>::value))
void decode_blocks(const void *__restrict__ blocks,
		uint8_t *__restrict__ pixels, uint32_t n_blocks, uint32_t mode_mask, uint32_t flags,
		uint8_t *__restrict__ ok_out, uint32_t *__restrict__ status) {
	constexpr int ROW = EpilogueOf<Dec, EPI>::kRowDwords;	// = 16-byte vectors per decoded block
	typedef uint32_t v4 __attribute__((ext_vector_type(4)));
	prepare_tables<Dec>();
	prepare_epilogue<Dec, EPI>();
	{
		const uint32_t i = blockIdx.x * 256u + threadIdx.x;
		const bool live = i < n_blocks;
		if constexpr (ROW == 1) {
			// 16 bytes per block: the wave's output is already one contiguous 1 KiB run per store instruction
			if (!live) return;
			uint32_t o[4];
			const bool ok = decode_block<Dec, EPI, CHECKED>(blocks, i, mode_mask, flags, o);
			store_with_policy<PolicyFor<Dec, Tune::kStorePolicyBlocks>::value, 4>(v4{ o[0], o[1], o[2], o[3] }, reinterpret_cast<v4 *>(pixels) + i);
			if (ok_out) ok_out[i] = ok ? 1 : 0;
			raise_status(!ok, status);
		} else {
			// A lane's block is ROW vectors = 16*ROW contiguous bytes, so a direct store instruction would write
			// 16 bytes out of every 16*ROW: partial lines, measured 91 us against 43 us for the linear layout on
			// BC1 8192^2.  The wave's blocks are contiguous in the output, so they go through LDS and are written
			// back as ROW instructions of one contiguous 1 KiB run each.  LDS layout [vector k of the block][block],
			// row stride GROUP + 16/ROW vectors: the writes (consecutive lanes, consecutive 16-byte slots) and the
			// transposed reads (output vector e = block*ROW + k) are both bank-conflict free; a lane-major layout
			// costs an 8-way conflict on every access for the 128-byte BC6H blocks.  64-bit pixels take two passes
			// of 32 blocks so that 17 KiB per workgroup suffice for every pixel size.  Lanes past the end of the
			// stream stay alive for the exchange (a tail wave's data is spread over all its lanes).
			// BC7 (64-byte blocks) stages inside its own, by then dead, per-lane rows instead (OwnStage).
			constexpr int PASSES = ROW == 8 ? 2 : 1, GROUP = 64 / PASSES;		// blocks per pass
			constexpr int STRIDE = GROUP + (16 % ROW == 0 ? 16 / ROW : 5);
			constexpr bool OWN = OwnStage<Dec>::value && ROW == 4;
			auto slot = [&](uint32_t k, uint32_t b) -> v4 * {
				if constexpr (OWN) {
					return static_cast<v4 *>(Dec::stage_slot(k, b));
				} else {
					__shared__ v4 stage[4][ROW * STRIDE];
					return &stage[threadIdx.x >> 6][k * STRIDE + b];
				}
			};
			const uint32_t lane = threadIdx.x & 63u;
			uint32_t o[4 * ROW];
			bool ok = true;
			if (live) ok = decode_block<Dec, EPI, CHECKED>(blocks, i, mode_mask, flags, o);
			const uint32_t first = i - lane;						// the wave's first block
			const uint32_t vectors = (first < n_blocks ? min(64u, n_blocks - first) : 0u) * ROW;
			v4 *out = reinterpret_cast<v4 *>(pixels) + (uint64_t)first * ROW;
			if constexpr (OWN) {
				// the staging slots are OTHER lanes' decoder rows: every lane of the wave must be done with its decode
				// (all of its row reads retired) before the first staging store lands there
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
				__builtin_amdgcn_wave_barrier();
			}
#pragma unroll
			for (int p = 0; p < PASSES; p++) {
				if (live && (PASSES == 1 || lane / GROUP == (uint32_t)p)) {
#pragma unroll
					for (int k = 0; k < ROW; k++) *slot((uint32_t)k, lane % GROUP) = v4{ o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3] };
				}
				// same wave: LDS operations complete in order; the fences keep the compiler from reordering across the exchange
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
				__builtin_amdgcn_wave_barrier();
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
				// A full wave -- every wave but the stream's last -- stores unguarded (`vectors` is wave-uniform: a scalar branch, no
				// compare and exec-mask update per store); decoders that own the staging slots give a lane's read address once, the
				// slots of blocks b + 16, b + 32, b + 48 a fixed step further on (Dec::stage_step).
				if (__builtin_amdgcn_readfirstlane((int)vectors) == 64 * ROW) {
					if constexpr (OWN) {
						const char *rd = static_cast<const char *>(Dec::stage_slot(lane % ROW, lane / ROW));
						const uint32_t step = Dec::stage_step(lane % ROW);
						v4 staged[GROUP * ROW / 64];			// all reads in flight before the first store (the stores are opaque to the scheduler)
#pragma unroll
						for (int j = 0; j < GROUP * ROW / 64; j++) staged[j] = *reinterpret_cast<const v4 *>(rd + (uint32_t)j * step);
#pragma unroll
						for (int j = 0; j < GROUP * ROW / 64; j++)
							store_with_policy<PolicyFor<Dec, Tune::kStorePolicyBlocks>::value, 4>(staged[j], out + (uint32_t)j * 64u + lane);
					} else {
						// (64-bit pixels keep read-store pairs: with all four reads of a pass in flight before its stores BC6H on coherent
						// content fell into its slow state -- 80.6 -> 85.9 us on the tiled fixture, same run; profiles/AB_RECORD.md)
						constexpr bool kReadsFirst = ROW != 8;
						v4 staged[GROUP * ROW / 64];
						if constexpr (kReadsFirst) {
#pragma unroll
							for (int j = 0; j < GROUP * ROW / 64; j++) { const uint32_t e = (uint32_t)j * 64u + lane; staged[j] = *slot(e % ROW, e / ROW); }
						}
#pragma unroll
						for (int j = 0; j < GROUP * ROW / 64; j++) {
							const uint32_t e = (uint32_t)j * 64u + lane;
							if constexpr (!kReadsFirst) staged[j] = *slot(e % ROW, e / ROW);
							store_with_policy<PolicyFor<Dec, Tune::kStorePolicyBlocks>::value, 4>(staged[j], out + (uint32_t)p * (GROUP * ROW) + e);
						}
					}
				} else {
#pragma unroll
					for (int j = 0; j < GROUP * ROW / 64; j++) {
						const uint32_t e = (uint32_t)j * 64u + lane;				// vector inside this pass
						const uint32_t g = (uint32_t)p * (GROUP * ROW) + e;			// vector inside the wave's output
						if (g < vectors) store_with_policy<PolicyFor<Dec, Tune::kStorePolicyBlocks>::value, 4>(*slot(e % ROW, e / ROW), out + g);
					}
				}
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
				__builtin_amdgcn_wave_barrier();
			}
			if (!live) return;
			if (ok_out) ok_out[i] = ok ? 1 : 0;
			raise_status(!ok, status);
		}
	}
}

}  // namespace detexhip
