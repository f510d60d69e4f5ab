// detexhip.hip -- C-ABI boundary of libdetexhip (include/detex.h + include/detexhip.h).
//
// Host side of the drop-in: the reference's link-time C API for the block-decode path
// (texture.c:55-145 drivers, the 19 leaf decoders of decompress-*.c, the misc.c:73-94 error
// convention) re-implemented as a thin C++ shim over the HIP kernels of kernels.h.  There is
// NO CPU decode in this library: every entry point, including the one-block leaf functions,
// runs the gfx950 kernels; without a usable HIP device the calls fail with an error message.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <chrono>
#include <cstring>
#include <mutex>
#include <thread>
#include <type_traits>

#define DETEXHIP_BUILDING_LIBRARY 1
#include "../../include/detex.h"
#include "../../include/detexhip.h"
#include "decode_s3tc_rgtc.h"
#include "decode_etc_eac.h"
#include "decode_bptc.h"
#include "decode_bptc_float.h"
#include "kernels.h"
#include "kernels_extra.h"
#ifdef DETEXHIP_AB_VARIANTS
// rejected A/B kernels: only in the measurement build (make lib-ab), never in the product library
#include "ab/variant_tile4x4.h"
#include "ab/decode_bptc_r01.h"
#include "ab/kernels_sorted.h"
#include "ab/kernels_persistent.h"
#endif

using namespace detexhip;

// ------------------------------------------------------------------------------------------------
// error convention (misc.c:73-94): thread-local, malloc'ed, replaced on every error
// ------------------------------------------------------------------------------------------------
static thread_local char *t_error_message = nullptr;

extern "C" void detexSetErrorMessage(const char *format, ...) {
	va_list args;
	va_start(args, format);
	char *message = nullptr;
	if (vasprintf(&message, format, args) < 0) message = strdup("detexSetErrorMessage: vasprintf returned error");
	va_end(args);
	free(t_error_message);
	t_error_message = message;
}

extern "C" const char *detexGetErrorMessage(void) { return t_error_message; }

#define HIP_TRY(expr, what)                                                                      \
	do {                                                                                         \
		hipError_t e_ = (expr);                                                                  \
		if (e_ != hipSuccess) {                                                                  \
			detexSetErrorMessage("libdetexhip: %s failed: %s", what, hipGetErrorString(e_));     \
			return false;                                                                        \
		}                                                                                        \
	} while (0)

// ------------------------------------------------------------------------------------------------
// format table: index = texture_format >> 24 (texture.c:27-48)
// ------------------------------------------------------------------------------------------------
namespace {

struct Geometry {
	const void *blocks; void *pixels; uint32_t wb, hb, width, height; uint64_t pitch;
	uint32_t *status; hipStream_t stream; int variant; int epi; uint32_t decode_flags;
	int resident;		// workgroups per CU the linear kernel of this format runs best with (FormatEntry::resident; 0 = no cap)
};
struct BatchArgs {
	const void *blocks; void *pixels; size_t n; uint32_t mode_mask, flags; uint8_t *ok; uint32_t *status;
	hipStream_t stream; bool checked; int epi; int resident;
};

// Calls fn(std::integral_constant<int, EPI>) for the epilogue `epi` if the decoder's native pixel class can feed it
// (kernels.h: RGBA8-class natives take the R<->B swap and the RGB8 packing; 1/2-component natives and unsigned BC6H
// the three "to 8-bit RGB(X)" epilogues; BC6H also its own R<->B swap); hipErrorInvalidValue otherwise.
template <class Dec, class F> hipError_t with_epilogue(int epi, F &&fn) {
	constexpr int NC = NativeOf<Dec>::value;
	if (epi == kEpiNone) return fn(std::integral_constant<int, kEpiNone>{});
	if constexpr (NC == kNatRGBA8) {
		if (epi == kEpiSwapRB8) return fn(std::integral_constant<int, kEpiSwapRB8>{});
		if (epi == kEpiPackRGB8) return fn(std::integral_constant<int, kEpiPackRGB8>{});
	} else if constexpr (NC != kNatOther) {
		if constexpr (NC == kNatFloatRGBX16) {
			if (epi == kEpiSwapRB16) return fn(std::integral_constant<int, kEpiSwapRB16>{});
		}
		if (epi == kEpiToRGBX8) return fn(std::integral_constant<int, kEpiToRGBX8>{});
		if (epi == kEpiToBGRX8) return fn(std::integral_constant<int, kEpiToBGRX8>{});
		if (epi == kEpiToRGB8) return fn(std::integral_constant<int, kEpiToRGB8>{});
	}
	return hipErrorInvalidValue;
}

// decoders whose throughput kernels carry wave-uniform specialisations use the plain form in the kernels that
// are not on the throughput path (clipped geometry, mip levels), to bound code size
template <class Dec> struct PlainDecoder { using type = Dec; };
template <> struct PlainDecoder<DecBPTCT<true>> { using type = DecBPTCPlain; };

#ifdef DETEXHIP_AB_VARIANTS
#include "ab/ab_dispatch.h"
constexpr int kMaxVariant = 7;
#else
constexpr int kMaxVariant = 0;
#endif

// decode_linear's geometry: whole blocks, vector-aligned rows.  `sector_aligned`: every wave's 1 KiB store run also starts on
// a 64-byte boundary (row bytes, pitch and base multiples of 64) -- where it does not, the staged kernel is the faster one.
template <class Dec, int EPI> bool fast_geometry(const Geometry &g, bool *sector_aligned) {
	constexpr unsigned piece = 4u * EpilogueOf<Dec, EPI>::kRowDwords;
	constexpr unsigned align = piece % 16u == 0 ? 16u : (piece % 8u == 0 ? 8u : 4u);
	const uintptr_t place = reinterpret_cast<uintptr_t>(g.pixels) | (uintptr_t)g.pitch;
	*sector_aligned = ((place | ((uintptr_t)g.wb * piece)) & 63u) == 0;
	return (g.width & 3u) == 0 && (g.height & 3u) == 0 && g.wb * 4u == g.width && g.hb * 4u == g.height && (place % align) == 0;
}

// Dynamic LDS to request at launch so that at most `target` workgroups of `kernel` are resident per CU (0 = no cap).  The
// linear kernels are store-bound, and the write path runs better with FEWER concurrent store streams than the eight workgroups
// per CU the registers allow (BC1 8192^2 42.5 -> 41.5 us at three to five per CU, BC6H on coherent content
// 92.6 -> 84.7 at three); unused LDS is the one launch-time handle on residency.  160 KiB per CU; the request is rounded down to
// 2 KiB so that `target` workgroups fit whatever the allocation granule, and target + 1 never do (true for 3 <= target <= 7).
template <auto Kernel> unsigned occupancy_cap_lds(int target) {		// (one cached attribute query per kernel instantiation)
	if (target < 3 || target > 7) return 0u;
	static const size_t static_lds = [] {
		hipFuncAttributes a{};
		return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(Kernel)) == hipSuccess ? a.sharedSizeBytes : (size_t)0;
	}();
	const size_t per_workgroup = ((size_t)163840 / (size_t)target) & ~(size_t)2047;
	return per_workgroup > static_lds ? (unsigned)(per_workgroup - static_lds) : 0u;
}
constexpr int workgroups_per_cu(int per_format) { return Tune::kWorkgroupsPerCu >= 0 ? Tune::kWorkgroupsPerCu : per_format; }

template <class Dec, int EPI> hipError_t launch_linear_epi(const Geometry &g) {
	const uint32_t n = g.wb * g.hb, tiles = (n + 255u) / 256u;
	uint8_t *px = static_cast<uint8_t *>(g.pixels);
	bool sector_aligned = false;
	// (formats with narrow rows keep the throughput kernels at any width: a 64-block run of R8 pixels is 256 bytes, and a
	// power-of-two width keeps those on sector boundaries anyway)
	constexpr bool kStagedWhenUnaligned = EpilogueOf<Dec, EPI>::kRowDwords >= 3;
	if (fast_geometry<Dec, EPI>(g, &sector_aligned) && (sector_aligned || !kStagedWhenUnaligned)) {
#ifdef DETEXHIP_AB_VARIANTS
		hipError_t ab_result;
		if (g.variant != 0 && ab_launch_linear<Dec, EPI>(g, &ab_result)) return ab_result;
#endif
		// narrow pixels (RGTC1, SIGNED_RGTC1): several blocks per lane, so that a store instruction covers a longer run
		constexpr int kRow = EpilogueOf<Dec, EPI>::kRowDwords, kGroup = kRow * LaneBlocks<Dec>::value <= 4 ? LaneBlocks<Dec>::value : 1;
		if constexpr (kGroup > 1) {
			if (g.wb % kGroup == 0 && (reinterpret_cast<uintptr_t>(px) | g.pitch) % (4u * kRow * kGroup) == 0) {
				constexpr auto kernel = &decode_linear_grouped<Dec, EPI, true, kGroup>;
				hipLaunchKernelGGL(kernel, dim3((n / kGroup + 255u) / 256u), dim3(256), occupancy_cap_lds<kernel>(workgroups_per_cu(g.resident)), g.stream,
					g.blocks, px, g.wb, n, g.pitch, g.status, g.decode_flags);
				return hipGetLastError();
			}
		}
		// non-temporal row stores (43 vs 51 us with cached stores on BC1 8192^2)
		constexpr auto kernel = &decode_linear<Dec, EPI, true>;
		hipLaunchKernelGGL(kernel, dim3(tiles), dim3(256), occupancy_cap_lds<kernel>(workgroups_per_cu(g.resident)), g.stream, g.blocks, px, g.wb, n, g.pitch,
			g.status, g.decode_flags);
		return hipGetLastError();
	}
	// Everything else whose rows are dword-aligned and a whole number of dwords long -- clipped sizes (texture.c:116-120,
	// 132-136), rows that do not start on 64-byte boundaries -- goes through the staged kernel (kernels.h), one launch; rows
	// that are not even dword-aligned (R8 / RG8 / RGB8 images of odd width) pixel by pixel.
	using Plain = typename PlainDecoder<Dec>::type;
	const size_t row_bytes = (size_t)g.width * (EpilogueOf<Dec, EPI>::kRowDwords);		// width pixels * (4 * ROW / 4) bytes
	if (((reinterpret_cast<uintptr_t>(px) | (uintptr_t)g.pitch | row_bytes) & 3u) == 0 && row_bytes <= 0xFFFFFFFFull) {
		const uint32_t tiles_per_row = (g.wb + 255u) / 256u;
		constexpr auto kernel = &decode_linear_staged<Plain, EPI>;
		hipLaunchKernelGGL(kernel, dim3(tiles_per_row * g.hb), dim3(256), occupancy_cap_lds<kernel>(workgroups_per_cu(g.resident)), g.stream, g.blocks, px, g.wb,
			(uint32_t)row_bytes, g.height, g.pitch, g.status, tiles_per_row, g.decode_flags);
	} else {
		hipLaunchKernelGGL((decode_linear_clipped<Plain, EPI>), dim3(tiles), dim3(256), 0, g.stream, g.blocks, px, g.wb, n, g.width, g.height, g.pitch,
			g.status, g.decode_flags);
	}
	return hipGetLastError();
}

template <class Dec> hipError_t launch_linear(const Geometry &g) {
	if (g.wb * g.hb == 0) return hipSuccess;
	return with_epilogue<Dec>(g.epi, [&](auto epi) { return launch_linear_epi<Dec, decltype(epi)::value>(g); });
}

template <class Dec, int EPI> hipError_t launch_blocks_epi(const BatchArgs &a) {
	const uint32_t tiles = (uint32_t)((a.n + 255u) / 256u);
	uint8_t *px = static_cast<uint8_t *>(a.pixels);
	if (a.checked) {
		hipLaunchKernelGGL((decode_blocks<typename PlainDecoder<Dec>::type, EPI, true>), dim3(tiles), dim3(256), 0, a.stream, a.blocks, px, (uint32_t)a.n,
			a.mode_mask, a.flags, a.ok, a.status);
	} else {
		constexpr auto kernel = &decode_blocks<Dec, EPI, false>;		// the block-major texture driver: store-bound like the linear kernel
		hipLaunchKernelGGL(kernel, dim3(tiles), dim3(256), occupancy_cap_lds<kernel>(workgroups_per_cu(a.resident)), a.stream, a.blocks, px, (uint32_t)a.n, a.mode_mask,
			a.flags, a.ok, a.status);
	}
	return hipGetLastError();
}

template <class Dec> hipError_t launch_blocks(const BatchArgs &a) {
	if (a.n == 0) return hipSuccess;
	return with_epilogue<Dec>(a.epi, [&](auto epi) { return launch_blocks_epi<Dec, decltype(epi)::value>(a); });
}

// one block handed over as a kernel argument (kernels_extra.h: decode_single)
struct SingleArgs { const uint8_t *bitstring; uint32_t mode_mask, flags; uint32_t *pixels; uint8_t *ok; hipStream_t stream; int epi; };
template <class Dec> hipError_t launch_single(const SingleArgs &a) {
	using Plain = typename PlainDecoder<Dec>::type;
	typename BlockWord<Dec::kBlockBytes>::type blk;
	memcpy(&blk, a.bitstring, sizeof blk);
	return with_epilogue<Dec>(a.epi, [&](auto epi) {
		hipLaunchKernelGGL((decode_single<Plain, decltype(epi)::value>), dim3(1), dim3(256), 0, a.stream, blk, a.mode_mask, a.flags, a.pixels, a.ok);
		return hipGetLastError();
	});
}

// 8f-3: all levels of a mip chain in one launch (kernels_extra.h)
struct LevelsArgs { LevelTable table; uint32_t *status; hipStream_t stream; int epi; uint32_t decode_flags; };
template <class Dec, int EPI> hipError_t launch_levels_epi(LevelsArgs &a) {
	constexpr unsigned row_bytes = 4u * EpilogueOf<Dec, EPI>::kRowDwords;
	constexpr unsigned align = row_bytes % 16u == 0 ? 16u : (row_bytes % 8u == 0 ? 8u : 4u);
	for (uint32_t l = 0; l < a.table.n_levels; l++) {
		LevelDesc &lv = a.table.level[l];
		const uint32_t hb = lv.width_in_blocks ? lv.n_blocks / lv.width_in_blocks : 0;
		lv.fast = (lv.width & 3u) == 0 && (lv.height & 3u) == 0 && lv.width_in_blocks * 4u == lv.width && hb * 4u == lv.height &&
			(reinterpret_cast<uintptr_t>(lv.pixels) % align) == 0 && (lv.pitch % align) == 0;
	}
	const uint32_t grid = a.table.wg_start[a.table.n_levels];
	if (grid == 0) return hipSuccess;
	hipLaunchKernelGGL((decode_levels<typename PlainDecoder<Dec>::type, EPI>), dim3(grid), dim3(256), 0, a.stream, a.table, a.status, a.decode_flags);
	return hipGetLastError();
}
template <class Dec> hipError_t launch_levels(LevelsArgs &a) {
	return with_epilogue<Dec>(a.epi, [&](auto epi) { return launch_levels_epi<Dec, decltype(epi)::value>(a); });
}

// 8f-4: block-mode histogram (kernels_extra.h)
template <int CLASS, int DWORDS> hipError_t launch_histogram(const void *blocks, size_t n, uint32_t *hist, hipStream_t stream, bool zero_first) {
	hipError_t e = zero_first ? hipMemsetAsync(hist, 0, 16 * sizeof(uint32_t), stream) : hipSuccess;
	if (e != hipSuccess || n == 0) return e;
	// 1024-lane workgroups, eight loads in flight per lane; every further workgroup adds serialised global atomics at the end
	// (kernels_extra.h).  Measured, 4 Mi / 16 Mi blocks, us per call incl. the memset: BC7 grid 96: 15.3, 128: 14.0 / 40.5,
	// 192: 14.0, 256: 14.3 / 42.2, 384: 14.7; ETC2 128: 11.8 / 32.3, 192: 10.6, 256: 10.4 / 24.4, 384: 11.4
	// (round 2 began at 22.0 and 18.0 with 256 workgroups of 256 lanes).
	const unsigned max_grid = DWORDS == 2 ? 256u : 192u;
	const size_t tiles = (n + kHistogramLanes - 1) / kHistogramLanes;
	const unsigned grid = (unsigned)(tiles < max_grid ? tiles : max_grid);
	hipLaunchKernelGGL((mode_histogram<CLASS, DWORDS>), dim3(grid), dim3(kHistogramLanes), 0, stream, static_cast<const uint32_t *>(blocks), (uint32_t)n, hist);
	return hipGetLastError();
}

struct FormatEntry {
	const char *name;
	uint32_t texture_format;
	hipError_t (*linear)(const Geometry &);
	hipError_t (*blocks)(const BatchArgs &);
	hipError_t (*single)(const SingleArgs &);
	hipError_t (*levels)(LevelsArgs &);
	hipError_t (*histogram)(const void *, size_t, uint32_t *, hipStream_t, bool);
	const char *kernel_name;
	int resident;		// resident workgroups per CU of the linear kernels (occupancy_cap_lds); 0 = whatever fits
	int resident_blocks;	// the same for the block-major texture driver
};

#define FMT(NAME, DEC, CLS, RESIDENT, RESIDENT_BLOCKS) { #NAME, DETEX_TEXTURE_FORMAT_##NAME, &launch_linear<DEC>, &launch_blocks<DEC>, &launch_single<DEC>, \
	&launch_levels<DEC>, &launch_histogram<CLS, DEC::kBlockBytes / 4>, "decode_linear<detexhip::" #DEC, RESIDENT, RESIDENT_BLOCKS }

// Last column: resident workgroups per CU of the linear kernel (occupancy_cap_lds), from the sweep of round 3
// (tools/gpu_wg_sweep.sh, profiles/r03/wg_sweep/; 8192^2, streams U / C, caps 3..7 against none).  The store-bound kernels with
// 32-bit or wider pixels and little VALU work gain 1-2.6 % at four or five workgroups per CU (BC1 42.1 -> 41.9, BC1A on its fixture
// 40.1 -> 39.3, BC3 47.5 -> 46.6, ETC1 42.8 -> 42.1); BC6H gains 11 % on coherent content (92.2 -> 82.0 us at five; its fixture
// tiled, which is what encoder-made textures look like) at the price of 2.6 % on uniform-random blocks, where the kernel sits at the
// board's power cap and needs every wave (85.2 -> 87.4) -- taken; signed BC6H has no fixture to show a gain and keeps all of them,
// as do BC7, ETC2_EAC and the narrow RGTC1 formats, which lose with any cap; ETC2 / punchthrough gain 3-4 % on their fixtures at
// five / six and nothing on random data.  The block-major driver (second number): the plain formats gain 1-2 % at five as well, ETC2
// and BC6H lose (their staging already takes the LDS of several workgroups) and keep what fits.
// (Re-swept with the `sc1 nt` row stores at the end of round 3, profiles/r03/explore_r03i/wg_sweep_sc1nt.jsonl: the table stands except for
// EAC_R11 / EAC_SIGNED_R11, which had six and now run best uncapped: 23.2 -> 22.7 us, on the fixture 22.5 -> 21.6, and ETC2, which takes
// `sc1 nt` only together with six instead of five: 42.3 / 42.0 -> 41.8 / 40.6.  Block-major, same re-sweep (tiled_resident_sc1nt.jsonl): the ETC
// family at seven per CU (ETC2 43.4 / 41.8 -> 42.0 / 40.8, ETC1, punchthrough and ETC2_EAC 1.3-1.7 %), BC6H at four: its fixture 91.7 -> 81.5 us
// at the price of 85.0 -> 89.0 on random blocks -- the same trade as in the linear kernel.)
const FormatEntry kFormats[20] = {
	{ nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0 },
	FMT(BC1, DecBC1, kClassS3TC, 5, 5), FMT(BC1A, DecBC1A, kClassS3TC, 5, 5), FMT(BC2, DecBC2, kClassS3TCat8, 5, 5), FMT(BC3, DecBC3, kClassS3TCat8, 5, 5),
	FMT(RGTC1, DecRGTC1, kClassNone, 0, 0), FMT(SIGNED_RGTC1, DecSignedRGTC1, kClassNone, 0, 0), FMT(RGTC2, DecRGTC2, kClassNone, 6, 0),
	FMT(SIGNED_RGTC2, DecSignedRGTC2, kClassNone, 5, 5),
	FMT(BPTC_FLOAT, DecBPTCFloat, kClassBPTCFloat, 5, 4), FMT(BPTC_SIGNED_FLOAT, DecBPTCSignedFloat, kClassBPTCFloat, 0, 0),
	FMT(BPTC, DecBPTC, kClassBPTC, 0, 0),
	FMT(ETC1, DecETC1, kClassETC1, 5, 7), FMT(ETC2, DecETC2, kClassETC2, 6, 7), FMT(ETC2_PUNCHTHROUGH, DecETC2Punchthrough, kClassETC2PT, 6, 7),
	FMT(ETC2_EAC, DecETC2EAC, kClassETC2at8, 0, 7),
	FMT(EAC_R11, DecEACR11, kClassNone, 0, 0), FMT(EAC_SIGNED_R11, DecEACSignedR11, kClassNone, 0, 0), FMT(EAC_RG11, DecEACRG11, kClassNone, 5, 5),
	FMT(EAC_SIGNED_RG11, DecEACSignedRG11, kClassNone, 5, 5),
};

const FormatEntry *lookup_format(uint32_t texture_format) {
	const uint32_t idx = texture_format >> 24;
	if (idx == 0 || idx >= 20) return nullptr;	// the reference indexes its table unchecked (SURVEY A-11)
	return kFormats[idx].texture_format == texture_format ? &kFormats[idx] : nullptr;
}

// Target pixel formats of the block-decode path and the epilogue that produces each (-1 = not offered): the native
// one, the RGBX8 <-> RGBA8 no-op edge (convert.c:768-769, 1087-1092), and -- converted inside the kernel with the exact
// result of the path detexConvertPixels takes (kernels.h) -- what the reference's callers request: BGRA8 / BGRX8
// (validate.c:204-209, detex-view.c:182), RGB8 (detex-convert.c:283-284) and RGBA8 / RGBX8, for every format the
// reference itself can convert (checked against the compiled reference: tools/make_goldens.py); FLOAT_BGRX16 for BC6H.
// Like the reference, the signed 16-bit formats have no path to BGRA8 and BPTC_SIGNED_FLOAT none to any 8-bit format.
enum : uint32_t { kPixelBGRA8 = 0x33C, kPixelBGRX8 = 0x328, kPixelRGB8 = 0x220, kPixelFloatBGRX16 = 0x2729 };
int epilogue_for(uint32_t texture_format, uint32_t pixel_format) {
	const uint32_t native = texture_format & DETEX_TEXTURE_FORMAT_PIXEL_FORMAT_MASK;
	if (pixel_format == native) return kEpiNone;
	const bool to_rgbx = pixel_format == DETEX_PIXEL_FORMAT_RGBA8 || pixel_format == DETEX_PIXEL_FORMAT_RGBX8;
	const bool to_bgrx = pixel_format == kPixelBGRA8 || pixel_format == kPixelBGRX8;
	if (native == DETEX_PIXEL_FORMAT_RGBA8 || native == DETEX_PIXEL_FORMAT_RGBX8)
		return to_rgbx ? kEpiNone : (to_bgrx ? kEpiSwapRB8 : (pixel_format == kPixelRGB8 ? kEpiPackRGB8 : -1));
	if (native == DETEX_PIXEL_FORMAT_FLOAT_RGBX16 && pixel_format == kPixelFloatBGRX16) return kEpiSwapRB16;
	const bool unsigned_small = native == DETEX_PIXEL_FORMAT_R8 || native == DETEX_PIXEL_FORMAT_RG8 || native == DETEX_PIXEL_FORMAT_R16 ||
		native == DETEX_PIXEL_FORMAT_RG16 || native == DETEX_PIXEL_FORMAT_FLOAT_RGBX16;
	const bool signed_small = native == DETEX_PIXEL_FORMAT_SIGNED_R16 || native == DETEX_PIXEL_FORMAT_SIGNED_RG16;
	if (unsigned_small || signed_small) {
		if (to_rgbx) return kEpiToRGBX8;
		if (to_bgrx) return (signed_small && pixel_format == kPixelBGRA8) ? -1 : kEpiToBGRX8;
		if (pixel_format == kPixelRGB8) return kEpiToRGB8;
	}
	return -1;
}
bool pixel_format_accepted(uint32_t texture_format, uint32_t pixel_format) { return epilogue_for(texture_format, pixel_format) >= 0; }

// The FLOAT_RGBX16 -> 8-bit epilogues look each half up in kHalfToU8 (kernels.h).  Entry = the reference's
// FLOAT_RGBX16 -> RGBX16 -> RGBX8 path: f = half as float (exact), clamped to 0..1 (detex.h:941-948); u16 =
// lrintf(f * 65535.0f + 0.5f) with the multiply, the add and the conversion all rounding DOWN (half-float.c:304-312 sets
// FE_DOWNWARD); u8 = (u16 + 127) * 255 / 65535 (convert.c:299-313).  The float operations are reproduced in double
// (products and sums of these operands are exact there) and rounded down to float by hand, so the table does not depend
// on this translation unit's floating-point environment.  tests/test_oracle_pin.py compares all 65536 entries' effect
// with the compiled reference.
float round_down_to_float(double v) {
	float r = (float)v;
	if ((double)r > v) r = nextafterf(r, -INFINITY);
	return r;
}
uint8_t half_to_u8_entry(uint32_t h) {
	const uint32_t sign = h >> 15, exponent = (h >> 10) & 31u, mantissa = h & 1023u;
	double f;
	if (exponent == 31u) f = 2.0;		// Inf clamps to 1, and so does every NaN in the reference build (gcc -Ofast; pinned exhaustively); BC6H never decodes to either
	else if (exponent == 0u) f = ldexp((double)mantissa, -24);
	else f = ldexp((double)(mantissa + 1024u), (int)exponent - 25);
	if (sign && !(exponent == 31u && mantissa)) f = -f;	// (a NaN of either sign converts like +Inf)
	const double clamped = f < 0.0 ? 0.0 : (f > 1.0 ? 1.0 : f);
	const float product = round_down_to_float(clamped * 65535.0);
	const float sum = round_down_to_float((double)product + 0.5);
	const uint32_t u16 = (uint32_t)floor((double)sum) & 0xFFFFu;
	return (uint8_t)component16_to_8(u16);
}
std::mutex g_half_table_mutex;
bool g_half_table_ready[64];
hipError_t ensure_half_table() {
	int dev = 0;
	hipError_t e = hipGetDevice(&dev);
	if (e != hipSuccess) return e;
	if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
	std::lock_guard<std::mutex> lock(g_half_table_mutex);
	if (g_half_table_ready[dev]) return hipSuccess;
	static uint8_t table[65536];
	static bool built = false;
	if (!built) { for (uint32_t h = 0; h < 65536u; h++) table[h] = half_to_u8_entry(h); built = true; }
	e = hipMemcpyToSymbol(HIP_SYMBOL(kHalfToU8), table, sizeof table, 0, hipMemcpyHostToDevice);
	if (e == hipSuccess) g_half_table_ready[dev] = true;
	return e;
}
// The device tier launches on the caller's stream and keeps per-device state (the half-float table): a stream of another
// device than the current one would have the table uploaded to the wrong GPU and the kernel read zeros.  Refused instead.
bool stream_on_current_device(hipStream_t stream, const char *who) {
	if (stream == nullptr) return true;		// the null stream is the current device's
	hipDevice_t sdev = -1;
	int cur = -1;
	if (hipStreamGetDevice(stream, &sdev) != hipSuccess || hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); return true; }	// cannot tell: as before
	if ((int)sdev == cur) return true;
	detexSetErrorMessage("%s: the stream belongs to device %d but device %d is current (hipSetDevice to the stream's device before calling)", who, (int)sdev, cur);
	return false;
}
// epilogue for a (format, target) pair that pixel_format_accepted() has admitted, with its device table in place
// (-2 + error message if the table upload failed)
int prepared_epilogue(uint32_t texture_format, uint32_t pixel_format) {
	const int epi = epilogue_for(texture_format, pixel_format);
	if ((texture_format & DETEX_TEXTURE_FORMAT_PIXEL_FORMAT_MASK) == DETEX_PIXEL_FORMAT_FLOAT_RGBX16 && epi >= kEpiToRGBX8) {
		hipError_t e = ensure_half_table();
		if (e != hipSuccess) { detexSetErrorMessage("libdetexhip: half-float table upload failed: %s", hipGetErrorString(e)); return -2; }
	}
	return epi;
}

// ------------------------------------------------------------------------------------------------
// per-thread device context of the host-pointer tier: a stream and grow-only device staging buffers that
// detexhipReleaseThreadResources() hands back
// ------------------------------------------------------------------------------------------------
struct ThreadContext {
	bool ready = false;
	int device = -1;
	int variant = -1;
	int quirks = -1;		// detexhipSetQuirks; -1 = not read yet (DETEXHIP_QUIRKS, default all)
	hipStream_t stream = nullptr;
	void *d_in = nullptr, *d_out = nullptr;
	size_t in_cap = 0, out_cap = 0;
	uint32_t *d_status = nullptr;	// [0] status word, [1..] ok bytes of the one-block calls / histogram bins
	// small calls: a pinned host buffer the kernels read blocks from and write pixels / status into directly (see direct_exchange)
	uint8_t *h_pin = nullptr, *d_pin = nullptr;
	size_t pin_cap = 0;
	void release() {
		if (!ready) return;
		int prev = -1;
		(void)hipGetDevice(&prev);
		(void)hipSetDevice(device);
		(void)hipStreamSynchronize(stream);
		(void)hipFree(d_in); (void)hipFree(d_out); (void)hipFree(d_status);
		if (h_pin) (void)hipHostFree(h_pin);
		(void)hipStreamDestroy(stream);
		d_in = d_out = nullptr; d_status = nullptr; in_cap = out_cap = 0;
		h_pin = d_pin = nullptr; pin_cap = 0;
		stream = nullptr;
		ready = false;
		if (prev >= 0) (void)hipSetDevice(prev);
	}
	~ThreadContext() { release(); }
};
thread_local ThreadContext t_ctx;
void release_shard_slots();	// the multi-device entries' per-thread slots (defined with them)

// The host tier always runs on the context's device, whatever device the calling thread has made current since
// (e.g. torch.cuda.set_device): entry points hold one of these for their duration and the caller's device is
// restored on return.
struct DeviceScope {
	int prev = -1;
	bool ok = true;
	explicit DeviceScope(int device) {
		if (hipGetDevice(&prev) != hipSuccess) prev = -1;
		if (prev != device) {
			hipError_t e = hipSetDevice(device);
			if (e != hipSuccess) { detexSetErrorMessage("libdetexhip: hipSetDevice(%d) failed: %s", device, hipGetErrorString(e)); ok = false; }
		}
	}
	~DeviceScope() { int now = -1; if (prev >= 0 && hipGetDevice(&now) == hipSuccess && now != prev) (void)hipSetDevice(prev); }
};

bool context_ready() {
	ThreadContext &c = t_ctx;
	if (c.ready) return true;
	int count = 0;
	hipError_t e = hipGetDeviceCount(&count);
	if (e != hipSuccess || count <= 0) {
		detexSetErrorMessage("libdetexhip: no usable HIP device (%s); this library has no CPU decode path",
			e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
		return false;
	}
	if (c.device < 0) {
		const char *env = getenv("DETEXHIP_DEVICE");
		c.device = env ? atoi(env) : 0;
	}
	DeviceScope scope(c.device);
	if (!scope.ok) return false;
	HIP_TRY(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking), "hipStreamCreate");
	HIP_TRY(hipMalloc(&c.d_status, 64), "hipMalloc(status)");
	c.ready = true;
	return true;
}

int current_variant() {
	ThreadContext &c = t_ctx;
	if (c.variant < 0) {
#ifdef DETEXHIP_AB_VARIANTS
		const char *env = getenv("DETEXHIP_VARIANT");	// measurement build only
		c.variant = env ? atoi(env) : 0;
#else
		c.variant = 0;
#endif
		if (c.variant < 0 || c.variant > kMaxVariant) c.variant = 0;
	}
	return c.variant;
}

// the reference's two BPTC quirks are reproduced unless switched off for the calling thread
// (detexhipSetQuirks, or DETEXHIP_QUIRKS in the environment when the thread first decodes); returns the decoders' spec flags
uint32_t current_spec_flags() {
	ThreadContext &c = t_ctx;
	if (c.quirks < 0) {
		const char *env = getenv("DETEXHIP_QUIRKS");
		c.quirks = env ? (int)(strtoul(env, nullptr, 0) & DETEXHIP_QUIRKS_REFERENCE) : (int)DETEXHIP_QUIRKS_REFERENCE;
	}
	return ((c.quirks & DETEXHIP_QUIRK_BC7_MODE6_PBIT) ? 0u : kFlagSpecBc7Mode6PBit) | ((c.quirks & DETEXHIP_QUIRK_BC6H_MODE12_BIT63) ? 0u : kFlagSpecBc6hMode12Bit63);
}

bool reserve(void **buf, size_t *cap, size_t need) {
	if (need <= *cap) return true;
	if (*buf) HIP_TRY(hipFree(*buf), "hipFree");
	*buf = nullptr; *cap = 0;
	const size_t rounded = (need + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
	HIP_TRY(hipMalloc(buf, rounded), "hipMalloc(staging)");
	*cap = rounded;
	return true;
}

// Small calls of the host tier (the one-block leaf functions; textures up to Tune::kHostDirectBytes of blocks + pixels): the
// blocks are placed in a pinned, device-visible host buffer and the kernel reads them from there and writes pixels, ok bytes
// and the status word back into it -- ONE launch and one stream synchronisation instead of memset + upload + launch + two
// downloads (five runtime calls that cost more than the kernel's PCIe traffic for a few KiB).  Layout of the buffer:
// [status word, ok byte: 256 B][blocks, 256-byte aligned][pixels, 256-byte aligned].
struct DirectExchange { uint8_t *h_base, *d_base; size_t in_off, out_off; };
bool direct_exchange(ThreadContext &c, size_t in_bytes, size_t out_bytes, DirectExchange *x) {
	const size_t in_off = 256, out_off = in_off + ((in_bytes + 255) & ~(size_t)255), need = out_off + ((out_bytes + 255) & ~(size_t)255);
	if (need > c.pin_cap) {
		if (c.h_pin) HIP_TRY(hipHostFree(c.h_pin), "hipHostFree");
		c.h_pin = c.d_pin = nullptr; c.pin_cap = 0;
		const size_t rounded = (need + 65535) & ~(size_t)65535;
		void *h = nullptr, *d = nullptr;
		HIP_TRY(hipHostMalloc(&h, rounded, hipHostMallocMapped), "hipHostMalloc");
		if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipHostFree(h); detexSetErrorMessage("libdetexhip: hipHostGetDevicePointer failed"); return false; }
		c.h_pin = static_cast<uint8_t *>(h); c.d_pin = static_cast<uint8_t *>(d); c.pin_cap = rounded;
	}
	*x = DirectExchange{ c.h_pin, c.d_pin, in_off, out_off };
	return true;
}

// shared by the 19 leaf functions and detexDecompressBlock: one block through the GPU.
// Returns 1 = decoded, 0 = the decoder returned false, -1 = HIP/runtime failure (message set).
int decode_one_block(const FormatEntry *f, const uint8_t *bitstring, uint32_t mode_mask, uint32_t flags,
		uint8_t *pixel_buffer, uint32_t pixel_format) {
	if (!context_ready()) return -1;
	ThreadContext &c = t_ctx;
	DeviceScope scope(c.device);
	if (!scope.ok) return -1;
	const size_t bs = detexGetCompressedBlockSize(f->texture_format);
	const size_t out_bytes = 16u * (size_t)detexGetPixelSize(pixel_format);
	DirectExchange x;
	if (!direct_exchange(c, bs, out_bytes, &x)) return -1;
	x.h_base[4] = 0;								// the ok byte
	auto run = [&]() -> bool {
		const int epi = prepared_epilogue(f->texture_format, pixel_format);
		if (epi == -2) return false;
		SingleArgs a{ bitstring, mode_mask, (flags & 0x3FFFFFFFu) | current_spec_flags(), reinterpret_cast<uint32_t *>(x.d_base + x.out_off), x.d_base + 4, c.stream, epi };
		HIP_TRY(f->single(a), "kernel launch");
		HIP_TRY(hipStreamSynchronize(c.stream), "hipStreamSynchronize");
		return true;
	};
	if (!run()) return -1;
	if (!x.h_base[4]) return 0;
	memcpy(pixel_buffer, x.h_base + x.out_off, out_bytes);
	return 1;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// extension tier: device pointers (include/detexhip.h)
// ------------------------------------------------------------------------------------------------
extern "C" int detexhipGetDeviceCount(void) {
	int n = 0;
	return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

extern "C" int detexhipSetDevice(int device) {
	if (t_ctx.ready && t_ctx.device != device) {	// checked BEFORE touching the current device
		detexSetErrorMessage("libdetexhip: detexhipSetDevice(%d) after this thread already used device %d "
			"(detexhipReleaseThreadResources() first)", device, t_ctx.device);
		return 1;
	}
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) {
		detexSetErrorMessage("libdetexhip: detexhipSetDevice(%d): no such device (%d present)", device, count);
		return 1;
	}
	t_ctx.device = device;
	return 0;
}

extern "C" void detexhipReleaseThreadResources(void) { t_ctx.release(); release_shard_slots(); }

extern "C" uint8_t detexhipHalfFloatToUNorm8(uint16_t half_bits) { return half_to_u8_entry(half_bits); }

extern "C" const char *detexhipVersion(void) { return "libdetexhip 0.2 (gfx950; detex v0.1.2 block-decode ABI)"; }

extern "C" void detexhipSetQuirks(uint32_t quirks) { t_ctx.quirks = (int)(quirks & DETEXHIP_QUIRKS_REFERENCE); }
extern "C" uint32_t detexhipGetQuirks(void) { (void)current_spec_flags(); return (uint32_t)t_ctx.quirks; }

extern "C" void detexhipSetKernelVariant(int variant) { t_ctx.variant = (variant >= 0 && variant <= kMaxVariant) ? variant : 0; }
extern "C" int detexhipGetKernelVariant(void) { return current_variant(); }

extern "C" const char *detexhipKernelName(uint32_t texture_format) {
	const FormatEntry *f = lookup_format(texture_format);
	if (!f) return nullptr;
	if (current_variant() == 1 && (texture_format >> 24) == 1) return "decode_linear_tile4x4";
	return f->kernel_name;
}

// 8f-3 device tier: up to 16 levels, one launch
extern "C" int detexhipDecompressLevelsLinearDevice(uint32_t texture_format, const detexhipLevel *levels, int n_levels,
		uint32_t pixel_format, void *stream, uint32_t *d_status) {
	const char *who = "detexhipDecompressLevelsLinearDevice";
	const FormatEntry *f = lookup_format(texture_format);
	if (!f) { detexSetErrorMessage("%s: 0x%08X is not a block-compressed format of this library", who, texture_format); return 1; }
	if (!stream_on_current_device(static_cast<hipStream_t>(stream), who)) return 1;
	const int epi = prepared_epilogue(texture_format, pixel_format);
	if (epi == -2) return 1;
	if (epi < 0) { detexSetErrorMessage("%s: pixel format 0x%08X is outside the block-decode path for format 0x%08X", who, pixel_format, texture_format); return 1; }
	if (n_levels < 0 || n_levels > kMaxLevels || (n_levels > 0 && !levels)) { detexSetErrorMessage("%s: 0..%d levels per call", who, kMaxLevels); return 1; }
	const size_t px = (size_t)detexGetPixelSize(pixel_format), palign = px == 3 ? 1 : (px < 4 ? px : 4);
	LevelsArgs a{};
	a.status = d_status; a.stream = static_cast<hipStream_t>(stream); a.epi = epi; a.decode_flags = current_spec_flags();
	a.table.n_levels = (uint32_t)n_levels;
	uint32_t wg = 0;
	for (int l = 0; l < n_levels; l++) {
		const detexhipLevel &s = levels[l];
		if (s.width < 0 || s.height < 0 || s.width_in_blocks < 0 || s.height_in_blocks < 0 || s.pitch_bytes < (size_t)s.width * px ||
				(s.pitch_bytes % palign) != 0 || (reinterpret_cast<uintptr_t>(s.d_pixels) % palign) != 0 ||
				(uint64_t)s.width_in_blocks * (uint64_t)s.height_in_blocks > 0x7FFFFF00ull) {
			detexSetErrorMessage("%s: bad geometry in level %d", who, l);
			return 1;
		}
		LevelDesc &d = a.table.level[l];
		d.blocks = s.d_blocks; d.pixels = static_cast<uint8_t *>(s.d_pixels); d.pitch = s.pitch_bytes;
		d.width_in_blocks = (uint32_t)s.width_in_blocks; d.n_blocks = (uint32_t)(s.width_in_blocks * s.height_in_blocks);
		d.width = (uint32_t)s.width; d.height = (uint32_t)s.height;
		a.table.wg_start[l] = wg;
		wg += (d.n_blocks + 255u) / 256u;
	}
	a.table.wg_start[n_levels] = wg;
	hipError_t e = f->levels(a);
	if (e != hipSuccess) { detexSetErrorMessage("libdetexhip: kernel launch failed: %s", hipGetErrorString(e)); return 1; }
	return 0;
}

// 8f-4 device tier
static int mode_histogram_device(uint32_t texture_format, const void *d_blocks, size_t n_blocks, uint32_t *d_hist, void *stream, bool zero_first);
extern "C" int detexhipModeHistogramDevice(uint32_t texture_format, const void *d_blocks, size_t n_blocks, uint32_t *d_hist,
		void *stream) {
	return mode_histogram_device(texture_format, d_blocks, n_blocks, d_hist, stream, true);
}
extern "C" int detexhipModeHistogramAccumulateDevice(uint32_t texture_format, const void *d_blocks, size_t n_blocks, uint32_t *d_hist,
		void *stream) {
	return mode_histogram_device(texture_format, d_blocks, n_blocks, d_hist, stream, false);
}
static int mode_histogram_device(uint32_t texture_format, const void *d_blocks, size_t n_blocks, uint32_t *d_hist, void *stream, bool zero_first) {
	const FormatEntry *f = lookup_format(texture_format);
	if (!f) { detexSetErrorMessage("detexhipModeHistogramDevice: 0x%08X is not a block-compressed format of this library", texture_format); return 1; }
	if (n_blocks > 0xFFFFFF00ull || !d_hist || reinterpret_cast<uintptr_t>(d_blocks) % detexGetCompressedBlockSize(texture_format) != 0) {
		detexSetErrorMessage("detexhipModeHistogramDevice: bad arguments (d_hist NULL, d_blocks not block-aligned, or too many blocks)");
		return 1;
	}
	hipError_t e = f->histogram(d_blocks, n_blocks, d_hist, static_cast<hipStream_t>(stream), zero_first);
	if (e != hipSuccess) { detexSetErrorMessage("libdetexhip: kernel launch failed: %s", hipGetErrorString(e)); return 1; }
	return 0;
}

extern "C" int detexhipDecompressTextureLinearDevice(uint32_t texture_format, const void *d_blocks, int width,
		int height, int width_in_blocks, int height_in_blocks, void *d_pixels, size_t pitch_bytes,
		uint32_t pixel_format, void *stream, uint32_t *d_status) {
	const FormatEntry *f = lookup_format(texture_format);
	if (!f) { detexSetErrorMessage("detexhipDecompressTextureLinearDevice: 0x%08X is not a block-compressed format of this library", texture_format); return 1; }
	if (!pixel_format_accepted(texture_format, pixel_format)) {
		detexSetErrorMessage("detexhipDecompressTextureLinearDevice: pixel format 0x%08X is outside the block-decode path for format 0x%08X", pixel_format, texture_format);
		return 1;
	}
	const size_t px = (size_t)detexGetPixelSize(pixel_format);
	const size_t palign = px == 3 ? 1 : (px < 4 ? px : 4);	// 24-bit pixels are byte-addressed, 64-bit ones dword-addressed
	if (width < 0 || height < 0 || width_in_blocks < 0 || height_in_blocks < 0 || pitch_bytes < (size_t)width * px ||
			(pitch_bytes % palign) != 0 || (reinterpret_cast<uintptr_t>(d_pixels) % palign) != 0 ||
			(uint64_t)width_in_blocks * (uint64_t)height_in_blocks > 0xFFFFFF00ull) {
		detexSetErrorMessage("detexhipDecompressTextureLinearDevice: bad geometry %dx%d (%dx%d blocks, pitch %zu)", width, height, width_in_blocks, height_in_blocks, pitch_bytes);
		return 1;
	}
	if (reinterpret_cast<uintptr_t>(d_blocks) % detexGetCompressedBlockSize(texture_format) != 0) {	// blocks are fetched with one 8/16-byte load each
		detexSetErrorMessage("detexhipDecompressTextureLinearDevice: d_blocks must be %d-byte aligned", (int)detexGetCompressedBlockSize(texture_format));
		return 1;
	}
	if (!stream_on_current_device(static_cast<hipStream_t>(stream), "detexhipDecompressTextureLinearDevice")) return 1;
	const int epi = prepared_epilogue(texture_format, pixel_format);
	if (epi == -2) return 1;
	Geometry g{ d_blocks, d_pixels, (uint32_t)width_in_blocks, (uint32_t)height_in_blocks, (uint32_t)width, (uint32_t)height,
		(uint64_t)pitch_bytes, d_status, static_cast<hipStream_t>(stream), current_variant(), epi, current_spec_flags(), f->resident };
	hipError_t e = f->linear(g);
	if (e != hipSuccess) { detexSetErrorMessage("libdetexhip: kernel launch failed: %s", hipGetErrorString(e)); return 1; }
	return 0;
}

static int blocks_device(const char *who, uint32_t texture_format, const void *d_blocks, size_t n_blocks, uint32_t mode_mask,
		uint32_t flags, void *d_pixels, uint8_t *d_ok, uint32_t *d_status, void *stream, bool checked, int epi) {
	const FormatEntry *f = lookup_format(texture_format);
	if (!f) { detexSetErrorMessage("%s: 0x%08X is not a block-compressed format of this library", who, texture_format); return 1; }
	if (n_blocks > 0xFFFFFF00ull) { detexSetErrorMessage("%s: too many blocks", who); return 1; }
	if (reinterpret_cast<uintptr_t>(d_blocks) % detexGetCompressedBlockSize(texture_format) != 0 || reinterpret_cast<uintptr_t>(d_pixels) % 16u != 0) {
		detexSetErrorMessage("%s: d_blocks must be %d-byte aligned and d_pixels 16-byte aligned (block-major output is written with 16-byte vector stores)", who,
			(int)detexGetCompressedBlockSize(texture_format));
		return 1;
	}
	// the reference's flags occupy bits 0-2 (detex.h:397-411); the spec switches ride in bits 30-31
	BatchArgs a{ d_blocks, d_pixels, n_blocks, mode_mask, (flags & 0x3FFFFFFFu) | current_spec_flags(), d_ok, d_status, static_cast<hipStream_t>(stream), checked, epi, f->resident_blocks };
	hipError_t e = f->blocks(a);
	if (e != hipSuccess) { detexSetErrorMessage("libdetexhip: kernel launch failed: %s", hipGetErrorString(e)); return 1; }
	return 0;
}

extern "C" int detexhipDecompressTextureTiledDevice(uint32_t texture_format, const void *d_blocks, int width_in_blocks,
		int height_in_blocks, void *d_pixels, uint32_t pixel_format, void *stream, uint32_t *d_status) {
	if (lookup_format(texture_format) && !pixel_format_accepted(texture_format, pixel_format)) {
		detexSetErrorMessage("detexhipDecompressTextureTiledDevice: pixel format 0x%08X is outside the block-decode path for format 0x%08X", pixel_format, texture_format);
		return 1;
	}
	if (width_in_blocks < 0 || height_in_blocks < 0) { detexSetErrorMessage("detexhipDecompressTextureTiledDevice: bad geometry"); return 1; }
	if (!stream_on_current_device(static_cast<hipStream_t>(stream), "detexhipDecompressTextureTiledDevice")) return 1;
	const int epi = lookup_format(texture_format) ? prepared_epilogue(texture_format, pixel_format) : kEpiNone;
	if (epi == -2) return 1;
	return blocks_device("detexhipDecompressTextureTiledDevice", texture_format, d_blocks,
		(size_t)width_in_blocks * (size_t)height_in_blocks, DETEX_MODE_MASK_ALL, 0, d_pixels, nullptr, d_status, stream, false, epi);
}

extern "C" int detexhipDecompressBlocksDevice(uint32_t texture_format, const void *d_blocks, size_t n_blocks, uint32_t mode_mask,
		uint32_t flags, void *d_pixels, uint8_t *d_ok, void *stream) {
	return blocks_device("detexhipDecompressBlocksDevice", texture_format, d_blocks, n_blocks, mode_mask, flags, d_pixels, d_ok,
		nullptr, stream, true, kEpiNone);
}

// ------------------------------------------------------------------------------------------------
// multi-device entry: one texture, N shards of block rows, one calling thread
// ------------------------------------------------------------------------------------------------
extern "C" int detexhipShardRows(int height_in_blocks, int n_shards, int shard, int *row0, int *row1) {
	if (n_shards <= 0 || shard < 0 || shard >= n_shards || height_in_blocks < 0 || !row0 || !row1) {
		detexSetErrorMessage("detexhipShardRows: bad arguments");
		return 1;
	}
	*row0 = (int)((int64_t)shard * height_in_blocks / n_shards);
	*row1 = (int)((int64_t)(shard + 1) * height_in_blocks / n_shards);
	return 0;
}

namespace {
// Per calling thread and shard index: stream, events, status word and grow-only staging buffers, created on first use and
// kept between calls (a shard keeps its device); detexhipReleaseThreadResources() hands them back.  Thread-local, so
// concurrent callers never share a slot and the entry points take no lock.
struct ShardSlot {
	int device = -1;
	hipStream_t stream = nullptr;
	hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
	uint32_t *d_status = nullptr;
	void *d_upload = nullptr; size_t upload_cap = 0;	// blocks uploaded from host_blocks
	void *d_band = nullptr; size_t band_cap = 0;		// decoded band of the host-output entry
	bool used = false;					// received work in the current call
	void destroy() {
		if (device < 0) return;
		int prev = -1;
		(void)hipGetDevice(&prev);
		if (hipSetDevice(device) == hipSuccess) {
			if (stream) { (void)hipStreamSynchronize(stream); (void)hipStreamDestroy(stream); }
			if (e0) (void)hipEventDestroy(e0);
			if (e1) (void)hipEventDestroy(e1);
			if (e2) (void)hipEventDestroy(e2);
			(void)hipFree(d_status); (void)hipFree(d_upload); (void)hipFree(d_band);
		}
		*this = ShardSlot{};
		if (prev >= 0) (void)hipSetDevice(prev);
	}
};
struct ShardSlots {
	ShardSlot slot[64];
	void release() { for (ShardSlot &sl : slot) sl.destroy(); }
	~ShardSlots() { release(); }
};
thread_local ShardSlots t_shards;

// makes `device` current and the slot usable on it; a slot that fails half-way is torn down completely
hipError_t prepare_slot(ShardSlot &sl, int device) {
	hipError_t e = hipSetDevice(device);
	if (e != hipSuccess) return e;
	if (sl.device == device && sl.stream) return hipSuccess;
	sl.destroy();							// the shard moved to another device (or was never set up)
	if ((e = hipSetDevice(device)) != hipSuccess) return e;
	sl.device = device;						// from here on destroy() releases whatever exists
	if ((e = hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking)) != hipSuccess || (e = hipEventCreate(&sl.e0)) != hipSuccess ||
			(e = hipEventCreate(&sl.e1)) != hipSuccess || (e = hipEventCreate(&sl.e2)) != hipSuccess ||
			(e = hipMalloc(&sl.d_status, 64)) != hipSuccess) {
		sl.destroy();
		(void)hipSetDevice(device);
		return e;
	}
	return hipSuccess;
}
hipError_t grow(void **buf, size_t *cap, size_t need) {		// on the current device
	if (need <= *cap) return hipSuccess;
	if (*buf) (void)hipFree(*buf);
	*buf = nullptr; *cap = 0;
	const size_t rounded = (need + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
	hipError_t e = hipMalloc(buf, rounded);
	if (e == hipSuccess) *cap = rounded;
	return e;
}

// peer access device -> peer, enabled once per process and pair (the call costs milliseconds; it used to run on every gather)
std::mutex g_peer_mutex;
uint64_t g_peer_enabled[64];
void enable_peer_once(int device, int peer) {			// `device` is current
	if (device == peer || device < 0 || device >= 64 || peer < 0 || peer >= 64) return;
	std::lock_guard<std::mutex> lock(g_peer_mutex);
	if (g_peer_enabled[device] >> peer & 1u) return;
	hipError_t e = hipDeviceEnablePeerAccess(peer, 0);
	if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();	// not fatal: the peer copy is then staged by the runtime
	g_peer_enabled[device] |= (uint64_t)1 << peer;
}

void release_shard_slots() { t_shards.release(); }
}  // namespace

extern "C" int detexhipDecompressTextureLinearMultiDevice(uint32_t texture_format, const void *host_blocks, int width, int height,
		int width_in_blocks, int height_in_blocks, size_t pitch_bytes, uint32_t pixel_format, detexhipShard *shards, int n_shards,
		int gather_device, void *d_gathered, float *decode_wall_ms, float *gather_wall_ms) {
	const char *who = "detexhipDecompressTextureLinearMultiDevice";
	const FormatEntry *f = lookup_format(texture_format);
	if (!f || !pixel_format_accepted(texture_format, pixel_format)) {
		detexSetErrorMessage("%s: format 0x%08X -> pixel format 0x%08X is outside the block-decode path of libdetexhip", who, texture_format, pixel_format);
		return 1;
	}
	if (!shards || n_shards < 1 || n_shards > 64 || width < 0 || height < 0 || width_in_blocks < 0 || height_in_blocks < 0 ||
			(gather_device >= 0 && !d_gathered)) {
		detexSetErrorMessage("%s: bad arguments (1..64 shards, non-negative geometry, d_gathered with gather_device)", who);
		return 1;
	}
	const size_t px = (size_t)detexGetPixelSize(pixel_format), bs = detexGetCompressedBlockSize(texture_format);
	const size_t pitch = pitch_bytes ? pitch_bytes : (size_t)width * px;
	const size_t wb = (size_t)width_in_blocks;
	if (pitch < (size_t)width * px) { detexSetErrorMessage("%s: pitch_bytes %zu is smaller than a row (%zu bytes)", who, pitch, (size_t)width * px); return 1; }
	int prev = -1;
	(void)hipGetDevice(&prev);
	int rc = 0;
	auto fail = [&](const char *what, hipError_t e) { detexSetErrorMessage("%s: %s failed: %s", who, what, hipGetErrorString(e)); rc = 1; };
	for (int g = 0; g < n_shards; g++) t_shards.slot[g].used = false;
	// per-shard rows, streams, status words, uploads, conversion tables (all before the timed region)
	for (int g = 0; g < n_shards && rc == 0; g++) {
		detexhipShard &sh = shards[g];
		(void)detexhipShardRows(height_in_blocks, n_shards, g, &sh.row0, &sh.row1);
		sh.decode_ms = 0.f; sh.invalid_blocks = 0;
		ShardSlot &sl = t_shards.slot[g];
		hipError_t e = prepare_slot(sl, sh.device);
		if (e != hipSuccess) { fail("device / stream setup", e); break; }
		sl.used = true;
		if (prepared_epilogue(texture_format, pixel_format) == -2) { rc = 1; break; }	// the half-float table of THIS device
		if ((e = hipMemsetAsync(sl.d_status, 0, 4, sl.stream)) != hipSuccess) { fail("hipMemsetAsync", e); break; }
		const size_t n = (size_t)(sh.row1 - sh.row0) * wb * bs;
		if (!sh.d_blocks) {
			if (!host_blocks) { detexSetErrorMessage("%s: shard %d has no d_blocks and host_blocks is NULL", who, g); rc = 1; break; }
			if ((e = grow(&sl.d_upload, &sl.upload_cap, n ? n : 16)) != hipSuccess) { fail("hipMalloc(blocks)", e); break; }
			if (n && (e = hipMemcpyAsync(sl.d_upload, static_cast<const uint8_t *>(host_blocks) + (size_t)sh.row0 * wb * bs, n, hipMemcpyHostToDevice,
					sl.stream)) != hipSuccess) { fail("hipMemcpyAsync(H2D)", e); break; }
		}
		if (gather_device >= 0) enable_peer_once(sh.device, gather_device);	// direct peer copies over xGMI where the topology allows
	}
	for (int g = 0; g < n_shards && rc == 0; g++) {		// uploads done: the timed region starts with idle devices
		hipError_t e = hipSetDevice(shards[g].device);
		if (e == hipSuccess) e = hipStreamSynchronize(t_shards.slot[g].stream);
		if (e != hipSuccess) fail("hipStreamSynchronize", e);
	}
	const auto t0 = std::chrono::steady_clock::now();
	for (int g = 0; g < n_shards && rc == 0; g++) {
		detexhipShard &sh = shards[g];
		ShardSlot &sl = t_shards.slot[g];
		hipError_t e = hipSetDevice(sh.device);
		if (e != hipSuccess) { fail("hipSetDevice", e); break; }
		const size_t y0 = (size_t)sh.row0 * 4u, y1 = ((size_t)sh.row1 * 4u < (size_t)height) ? (size_t)sh.row1 * 4u : (size_t)height;
		(void)hipEventRecord(sl.e0, sl.stream);
		if (y1 > y0 && sh.row1 > sh.row0) {
			if (detexhipDecompressTextureLinearDevice(texture_format, sh.d_blocks ? sh.d_blocks : sl.d_upload, width, (int)(y1 - y0), width_in_blocks,
					sh.row1 - sh.row0, sh.d_pixels, pitch, pixel_format, sl.stream, sl.d_status) != 0) { rc = 1; break; }
		}
		(void)hipEventRecord(sl.e1, sl.stream);
		if (gather_device >= 0 && y1 > y0) {
			// a band is (rows - 1) * pitch + width * px bytes: one flat peer copy when rows are dense, a 2-D copy otherwise (the
			// bytes between width * px and pitch belong to the caller, in the band and in the gathered image alike)
			uint8_t *dst = static_cast<uint8_t *>(d_gathered) + y0 * pitch;
			if (pitch == (size_t)width * px) e = hipMemcpyPeerAsync(dst, gather_device, sh.d_pixels, sh.device, (y1 - y0) * pitch, sl.stream);
			else e = hipMemcpy2DAsync(dst, pitch, sh.d_pixels, pitch, (size_t)width * px, y1 - y0, hipMemcpyDeviceToDevice, sl.stream);
			if (e != hipSuccess) { fail("peer copy", e); break; }
		}
		(void)hipEventRecord(sl.e2, sl.stream);
	}
	if (rc == 0) {
		for (int g = 0; g < n_shards; g++) { (void)hipSetDevice(shards[g].device); hipError_t e = hipEventSynchronize(t_shards.slot[g].e1); if (e != hipSuccess && rc == 0) fail("kernel", e); }
		const auto t1 = std::chrono::steady_clock::now();
		for (int g = 0; g < n_shards; g++) { (void)hipSetDevice(shards[g].device); hipError_t e = hipEventSynchronize(t_shards.slot[g].e2); if (e != hipSuccess && rc == 0) fail("gather", e); }
		const auto t2 = std::chrono::steady_clock::now();
		if (decode_wall_ms) *decode_wall_ms = std::chrono::duration<float, std::milli>(t1 - t0).count();
		if (gather_wall_ms) *gather_wall_ms = gather_device >= 0 ? std::chrono::duration<float, std::milli>(t2 - t0).count() : 0.f;
		for (int g = 0; g < n_shards && rc == 0; g++) {
			(void)hipSetDevice(shards[g].device);
			uint32_t st = 0;
			hipError_t e = hipMemcpy(&st, t_shards.slot[g].d_status, 4, hipMemcpyDeviceToHost);
			if (e != hipSuccess) { fail("hipMemcpy(status)", e); break; }
			shards[g].invalid_blocks = st != 0;
			(void)hipEventElapsedTime(&shards[g].decode_ms, t_shards.slot[g].e0, t_shards.slot[g].e1);
		}
	}
	// Nothing launched by this call may still be running when it returns, failed or not: the caller is free to release its
	// buffers.  (The error message of the first failure stays.)
	for (int g = 0; g < n_shards; g++) {
		ShardSlot &sl = t_shards.slot[g];
		if (sl.used && sl.stream && hipSetDevice(sl.device) == hipSuccess) (void)hipStreamSynchronize(sl.stream);
		sl.used = false;
	}
	if (prev >= 0) (void)hipSetDevice(prev);
	return rc;
}

// Host image in, host image out, over N devices: every shard uploads its band of blocks, decodes it and downloads its band
// of pixels over ITS OWN PCIe link -- the one lever left for the host-pointer tier, which a single link bounds at ~56 GB/s
// (8192^2 RGBA8: 4.8 ms of download against 0.04 ms of kernel).  One worker thread per shard (copies
// from and to pageable memory block their caller); the workers use the calling thread's slots, which it does not touch
// until they have joined.
extern "C" int detexhipDecompressTextureLinearMultiDeviceHost(uint32_t texture_format, const void *host_blocks, int width, int height,
		int width_in_blocks, int height_in_blocks, void *host_pixels, size_t pitch_bytes, uint32_t pixel_format, const int *devices, int n_shards,
		int *any_invalid, float *wall_ms) {
	const char *who = "detexhipDecompressTextureLinearMultiDeviceHost";
	const FormatEntry *f = lookup_format(texture_format);
	if (!f || !pixel_format_accepted(texture_format, pixel_format)) {
		detexSetErrorMessage("%s: format 0x%08X -> pixel format 0x%08X is outside the block-decode path of libdetexhip", who, texture_format, pixel_format);
		return 1;
	}
	if (!devices || n_shards < 1 || n_shards > 64 || width < 0 || height < 0 || width_in_blocks < 0 || height_in_blocks < 0 || !host_blocks || !host_pixels) {
		detexSetErrorMessage("%s: bad arguments (1..64 shards, non-negative geometry, host_blocks and host_pixels)", who);
		return 1;
	}
	const size_t px = (size_t)detexGetPixelSize(pixel_format), bs = detexGetCompressedBlockSize(texture_format);
	const size_t pitch = pitch_bytes ? pitch_bytes : (size_t)width * px, row_bytes = (size_t)width * px, wb = (size_t)width_in_blocks;
	if (pitch < row_bytes) { detexSetErrorMessage("%s: pitch_bytes %zu is smaller than a row (%zu bytes)", who, pitch, row_bytes); return 1; }
	// like the reference, only the pixels the block grid covers are written (texture.c:116-136)
	const size_t cov_w = (size_t)width < 4u * wb ? (size_t)width : 4u * wb;
	int prev = -1;
	(void)hipGetDevice(&prev);
	struct Work { int rc = 0; bool invalid = false; char message[256] = { 0 }; };
	Work work[64];
	ShardSlots &slots = t_shards;
	const auto t0 = std::chrono::steady_clock::now();
	auto run = [&](int g) {
		Work &w = work[g];
		auto fail = [&](const char *what, hipError_t e) { snprintf(w.message, sizeof w.message, "%s: shard %d: %s failed: %s", who, g, what, hipGetErrorString(e)); w.rc = 1; };
		int row0 = 0, row1 = 0;
		(void)detexhipShardRows(height_in_blocks, n_shards, g, &row0, &row1);
		const size_t y0 = (size_t)row0 * 4u, y1 = ((size_t)row1 * 4u < (size_t)height) ? (size_t)row1 * 4u : (size_t)height;
		if (row1 <= row0 || y1 <= y0 || cov_w == 0) return;
		ShardSlot &sl = slots.slot[g];
		hipError_t e = prepare_slot(sl, devices[g]);
		if (e != hipSuccess) { fail("device / stream setup", e); return; }
		if (prepared_epilogue(texture_format, pixel_format) == -2) { snprintf(w.message, sizeof w.message, "%s: shard %d: conversion table upload failed", who, g); w.rc = 1; return; }
		const size_t n_in = (size_t)(row1 - row0) * wb * bs, rows = y1 - y0;
		if ((e = grow(&sl.d_upload, &sl.upload_cap, n_in)) != hipSuccess || (e = grow(&sl.d_band, &sl.band_cap, rows * row_bytes)) != hipSuccess) { fail("hipMalloc", e); return; }
		if ((e = hipMemsetAsync(sl.d_status, 0, 4, sl.stream)) != hipSuccess) { fail("hipMemsetAsync", e); return; }
		if ((e = hipMemcpyAsync(sl.d_upload, static_cast<const uint8_t *>(host_blocks) + (size_t)row0 * wb * bs, n_in, hipMemcpyHostToDevice, sl.stream)) != hipSuccess) { fail("hipMemcpyAsync(H2D)", e); return; }
		if (detexhipDecompressTextureLinearDevice(texture_format, sl.d_upload, width, (int)rows, width_in_blocks, row1 - row0, sl.d_band, row_bytes, pixel_format,
				sl.stream, sl.d_status) != 0) { snprintf(w.message, sizeof w.message, "%s", detexGetErrorMessage() ? detexGetErrorMessage() : "launch failed"); w.rc = 1; }
		uint8_t *dst = static_cast<uint8_t *>(host_pixels) + y0 * pitch;
		uint32_t st = 0;
		if (w.rc == 0) {
			if (pitch == row_bytes && cov_w == (size_t)width) e = hipMemcpyAsync(dst, sl.d_band, rows * row_bytes, hipMemcpyDeviceToHost, sl.stream);
			else e = hipMemcpy2DAsync(dst, pitch, sl.d_band, row_bytes, cov_w * px, rows, hipMemcpyDeviceToHost, sl.stream);
			if (e != hipSuccess) fail("download", e);
			else if ((e = hipMemcpyAsync(&st, sl.d_status, 4, hipMemcpyDeviceToHost, sl.stream)) != hipSuccess) fail("hipMemcpyAsync(status)", e);
		}
		if ((e = hipStreamSynchronize(sl.stream)) != hipSuccess && w.rc == 0) fail("hipStreamSynchronize", e);
		w.invalid = st != 0;
	};
	if (n_shards == 1) run(0);
	else {
		std::thread workers[64];
		for (int g = 0; g < n_shards; g++) workers[g] = std::thread(run, g);
		for (int g = 0; g < n_shards; g++) workers[g].join();
	}
	if (wall_ms) *wall_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
	if (prev >= 0) (void)hipSetDevice(prev);
	bool invalid = false;
	for (int g = 0; g < n_shards; g++) {
		if (work[g].rc != 0) { detexSetErrorMessage("%s", work[g].message); return 1; }
		invalid = invalid || work[g].invalid;
	}
	if (any_invalid) *any_invalid = invalid ? 1 : 0;
	if (invalid) detexSetErrorMessage("detexDecompressBlock: Decompress function for format 0x%08X returned error", texture_format);
	return 0;
}

// ------------------------------------------------------------------------------------------------
// reference tier: host pointers (include/detex.h)
// ------------------------------------------------------------------------------------------------
#define LEAF(NAME)                                                                                          \
	extern "C" bool detexDecompressBlock##NAME(const uint8_t *bitstring, uint32_t mode_mask, uint32_t flags, \
			uint8_t *pixel_buffer) {                                                                         \
		return decode_one_block(&kFormats[DETEX_TEXTURE_FORMAT_##NAME >> 24], bitstring, mode_mask, flags,  \
			pixel_buffer, DETEX_TEXTURE_FORMAT_##NAME & 0xFFFFu) == 1;                                       \
	}
LEAF(BC1) LEAF(BC1A) LEAF(BC2) LEAF(BC3) LEAF(RGTC1) LEAF(SIGNED_RGTC1) LEAF(RGTC2) LEAF(SIGNED_RGTC2)
LEAF(BPTC_FLOAT) LEAF(BPTC_SIGNED_FLOAT) LEAF(BPTC) LEAF(ETC1) LEAF(ETC2) LEAF(ETC2_PUNCHTHROUGH) LEAF(ETC2_EAC)
LEAF(EAC_R11) LEAF(EAC_SIGNED_R11) LEAF(EAC_RG11) LEAF(EAC_SIGNED_RG11)
#undef LEAF

// texture.c:55-70
extern "C" bool detexDecompressBlock(const uint8_t *bitstring, uint32_t texture_format, uint32_t mode_mask, uint32_t flags,
		uint8_t *pixel_buffer, uint32_t pixel_format) {
	const FormatEntry *f = lookup_format(texture_format);
	if (!f) {
		detexSetErrorMessage("detexDecompressBlock: 0x%08X is not a block-compressed format of this library", texture_format);
		return false;
	}
	if (!pixel_format_accepted(texture_format, pixel_format)) {
		detexSetErrorMessage("detexDecompressBlock: conversion of format 0x%08X to pixel format 0x%08X is outside the "
			"block-decode path of libdetexhip", texture_format, pixel_format);
		return false;
	}
	const int r = decode_one_block(f, bitstring, mode_mask, flags, pixel_buffer, pixel_format);
	if (r == 0)	// same text as the reference (texture.c:63-64); HIP failures have set their own message
		detexSetErrorMessage("detexDecompressBlock: Decompress function for format 0x%08X returned error", texture_format);
	return r == 1;
}

// shared body of the two texture drivers (texture.c:77-98, 105-145)
//
// upload -> one launch -> download on the thread's stream.  The PCIe download of the pixels bounds this tier (8192^2
// BC1: 256 MiB at the 56 GB/s pageable copies reach on the test box = 4.8 ms; upload 0.6 ms, kernel 0.04 ms: 5.4 ms).
// A band pipeline (upload k+1 | kernel k | download k-1 on three streams, uploads from a helper thread because
// hipMemcpyAsync on pageable memory blocks its caller) was built and measured: 5.40 vs 5.45 ms -- a download that
// shares the link with an upload runs at 41-50 GB/s instead of 56 and every extra copy call costs 30-50 us
// (tools/ubench/host_paths.hip) -- so it was not kept.
static bool decompress_texture(const detexTexture *texture, uint8_t *pixel_buffer, uint32_t pixel_format, bool tiled) {
	const char *who = tiled ? "detexDecompressTextureTiled" : "detexDecompressTextureLinear";
	const size_t px = (size_t)detexGetPixelSize(pixel_format);
	if (texture->width < 0 || texture->height < 0 || texture->width_in_blocks < 0 || texture->height_in_blocks < 0) {
		detexSetErrorMessage("%s: negative texture dimensions", who);
		return false;
	}
	const size_t wb = (size_t)texture->width_in_blocks, hb = (size_t)texture->height_in_blocks;
	const size_t width = (size_t)texture->width, height = (size_t)texture->height;
	const size_t out_bytes = tiled ? wb * hb * 16u * px : width * height * px;
	if (!detexFormatIsCompressed(texture->format)) {
		if (tiled) { detexSetErrorMessage("detexDecompressTextureTiled: Cannot handle uncompressed texture format"); return false; }
		// texture.c:108-111 hands uncompressed textures to detexConvertPixels; only its identity
		// edge (convert.c:1087-1092) belongs to this path.
		const uint32_t src = detexGetPixelFormat(texture->format);
		const bool same8 = (src == DETEX_PIXEL_FORMAT_RGBA8 || src == DETEX_PIXEL_FORMAT_RGBX8) &&
			(pixel_format == DETEX_PIXEL_FORMAT_RGBA8 || pixel_format == DETEX_PIXEL_FORMAT_RGBX8);
		if (src == pixel_format || same8) { memcpy(pixel_buffer, texture->data, out_bytes); return true; }
		detexSetErrorMessage("%s: pixel conversion 0x%08X -> 0x%08X is outside the block-decode path of libdetexhip", who, src, pixel_format);
		return false;
	}
	const FormatEntry *f = lookup_format(texture->format);
	if (!f || !pixel_format_accepted(texture->format, pixel_format)) {
		// the reference fails every block here: all-zero image and false
		memset(pixel_buffer, 0, out_bytes);
		detexSetErrorMessage("%s: format 0x%08X -> pixel format 0x%08X is outside the block-decode path of libdetexhip", who,
			texture->format, pixel_format);
		return false;
	}
	if (out_bytes == 0 || wb * hb == 0) return true;
	if (!context_ready()) return false;
	ThreadContext &c = t_ctx;
	DeviceScope scope(c.device);
	if (!scope.ok) return false;
	const size_t bs = detexGetCompressedBlockSize(texture->format);
	const size_t in_bytes = wb * hb * bs;
	// The reference writes only the pixels its block grid covers and the image contains (texture.c:116-136): when the
	// grid is smaller than the image, the rest of the caller's buffer is left untouched, not overwritten with staging bytes.
	const size_t cov_w = tiled ? 0 : (width < 4u * wb ? width : 4u * wb), cov_h = tiled ? 0 : (height < 4u * hb ? height : 4u * hb);
	if (in_bytes + out_bytes <= Tune::kHostDirectBytes) {
		// small texture: the kernel reads the blocks from, and writes pixels and status into, pinned host memory (direct_exchange)
		DirectExchange x;
		if (!direct_exchange(c, in_bytes, out_bytes, &x)) return false;
		memcpy(x.h_base + x.in_off, texture->data, in_bytes);
		*reinterpret_cast<volatile uint32_t *>(x.h_base) = 0;
		uint32_t *d_st = reinterpret_cast<uint32_t *>(x.d_base);
		int rc;
		if (tiled)
			rc = detexhipDecompressTextureTiledDevice(texture->format, x.d_base + x.in_off, (int)wb, (int)hb, x.d_base + x.out_off, pixel_format, c.stream, d_st);
		else
			rc = detexhipDecompressTextureLinearDevice(texture->format, x.d_base + x.in_off, (int)width, (int)height, (int)wb, (int)hb, x.d_base + x.out_off,
				width * px, pixel_format, c.stream, d_st);
		if (rc != 0) return false;
		HIP_TRY(hipStreamSynchronize(c.stream), "hipStreamSynchronize");
		const uint8_t *res = x.h_base + x.out_off;
		if (tiled || (cov_w == width && cov_h == height)) memcpy(pixel_buffer, res, out_bytes);
		else for (size_t y = 0; y < cov_h; y++) memcpy(pixel_buffer + y * width * px, res + y * width * px, cov_w * px);
		if (*reinterpret_cast<volatile uint32_t *>(x.h_base) != 0) {
			detexSetErrorMessage("detexDecompressBlock: Decompress function for format 0x%08X returned error", texture->format);
			return false;
		}
		return true;
	}
	if (!reserve(&c.d_in, &c.in_cap, in_bytes) || !reserve(&c.d_out, &c.out_cap, out_bytes)) return false;
	HIP_TRY(hipMemsetAsync(c.d_status, 0, 4, c.stream), "hipMemsetAsync");
	uint8_t *d_in = static_cast<uint8_t *>(c.d_in), *d_out = static_cast<uint8_t *>(c.d_out);
	uint32_t status = 0;
	HIP_TRY(hipMemcpyAsync(d_in, texture->data, in_bytes, hipMemcpyHostToDevice, c.stream), "hipMemcpyAsync(H2D)");
	int rc;
	if (tiled)
		rc = detexhipDecompressTextureTiledDevice(texture->format, d_in, (int)wb, (int)hb, d_out, pixel_format, c.stream, c.d_status);
	else
		rc = detexhipDecompressTextureLinearDevice(texture->format, d_in, (int)width, (int)height, (int)wb, (int)hb, d_out, width * px, pixel_format,
			c.stream, c.d_status);
	if (rc != 0) return false;
	if (tiled || (cov_w == width && cov_h == height)) {
		HIP_TRY(hipMemcpyAsync(pixel_buffer, d_out, out_bytes, hipMemcpyDeviceToHost, c.stream), "hipMemcpyAsync(D2H)");
	} else if (cov_w > 0 && cov_h > 0) {
		HIP_TRY(hipMemcpy2DAsync(pixel_buffer, width * px, d_out, width * px, cov_w * px, cov_h, hipMemcpyDeviceToHost, c.stream), "hipMemcpy2DAsync(D2H)");
	}
	HIP_TRY(hipMemcpyAsync(&status, c.d_status, 4, hipMemcpyDeviceToHost, c.stream), "hipMemcpyAsync(D2H)");
	HIP_TRY(hipStreamSynchronize(c.stream), "hipStreamSynchronize");
	if (status != 0) {
		// same text the reference leaves behind after a failed block (texture.c:63-64)
		detexSetErrorMessage("detexDecompressBlock: Decompress function for format 0x%08X returned error", texture->format);
		return false;
	}
	return true;
}

// 8f-3 host tier: what a caller of detexLoadKTXFileWithMipmaps does level by level (one
// detexDecompressTextureLinear per level), as one staging copy in, ONE launch, one copy out.
extern "C" bool detexhipDecompressTexturesLinear(const detexTexture *const *textures, int n_textures,
		uint8_t *const *pixel_buffers, uint32_t pixel_format) {
	const char *who = "detexhipDecompressTexturesLinear";
	if (n_textures <= 0) return true;
	if (n_textures > kMaxLevels) { detexSetErrorMessage("%s: at most %d textures per call", who, kMaxLevels); return false; }
	const uint32_t format = textures[0]->format;
	const FormatEntry *f = lookup_format(format);
	const size_t px = (size_t)detexGetPixelSize(pixel_format);
	if (!f || !pixel_format_accepted(format, pixel_format)) {
		for (int l = 0; l < n_textures; l++) memset(pixel_buffers[l], 0, (size_t)textures[l]->width * (size_t)textures[l]->height * px);
		detexSetErrorMessage("%s: format 0x%08X -> pixel format 0x%08X is outside the block-decode path of libdetexhip", who, format, pixel_format);
		return false;
	}
	if (!context_ready()) return false;
	ThreadContext &c = t_ctx;
	DeviceScope scope(c.device);
	if (!scope.ok) return false;
	const size_t bs = detexGetCompressedBlockSize(format);
	size_t in_off[kMaxLevels], out_off[kMaxLevels], in_total = 0, out_total = 0;
	for (int l = 0; l < n_textures; l++) {
		if (textures[l]->format != format) { detexSetErrorMessage("%s: all textures must share one format", who); return false; }
		in_off[l] = in_total; out_off[l] = out_total;
		in_total += ((size_t)textures[l]->width_in_blocks * (size_t)textures[l]->height_in_blocks * bs + 255) & ~(size_t)255;
		out_total += ((size_t)textures[l]->width * (size_t)textures[l]->height * px + 255) & ~(size_t)255;
	}
	if (!reserve(&c.d_in, &c.in_cap, in_total ? in_total : 256) || !reserve(&c.d_out, &c.out_cap, out_total ? out_total : 256)) return false;
	detexhipLevel lv[kMaxLevels];
	HIP_TRY(hipMemsetAsync(c.d_status, 0, 4, c.stream), "hipMemsetAsync");
	for (int l = 0; l < n_textures; l++) {
		const detexTexture *t = textures[l];
		const size_t nbytes = (size_t)t->width_in_blocks * (size_t)t->height_in_blocks * bs;
		if (nbytes) HIP_TRY(hipMemcpyAsync(static_cast<uint8_t *>(c.d_in) + in_off[l], t->data, nbytes, hipMemcpyHostToDevice, c.stream), "hipMemcpyAsync(H2D)");
		lv[l] = detexhipLevel{ static_cast<uint8_t *>(c.d_in) + in_off[l], static_cast<uint8_t *>(c.d_out) + out_off[l],
			(size_t)t->width * px, t->width, t->height, t->width_in_blocks, t->height_in_blocks };
	}
	if (detexhipDecompressLevelsLinearDevice(format, lv, n_textures, pixel_format, c.stream, c.d_status) != 0) return false;
	uint32_t status = 0;
	for (int l = 0; l < n_textures; l++) {
		const size_t nbytes = (size_t)textures[l]->width * (size_t)textures[l]->height * px;
		if (nbytes) HIP_TRY(hipMemcpyAsync(pixel_buffers[l], static_cast<uint8_t *>(c.d_out) + out_off[l], nbytes, hipMemcpyDeviceToHost, c.stream), "hipMemcpyAsync(D2H)");
	}
	HIP_TRY(hipMemcpyAsync(&status, c.d_status, 4, hipMemcpyDeviceToHost, c.stream), "hipMemcpyAsync(D2H)");
	HIP_TRY(hipStreamSynchronize(c.stream), "hipStreamSynchronize");
	if (status != 0) {
		detexSetErrorMessage("detexDecompressBlock: Decompress function for format 0x%08X returned error", format);
		return false;
	}
	return true;
}

// 8f-4 host tier: histogram[m] = number of blocks whose detexGetMode<FMT> is m (bin 15: reserved codes)
extern "C" bool detexhipModeHistogram(uint32_t texture_format, const uint8_t *blocks, size_t n_blocks, uint32_t histogram[16]) {
	const FormatEntry *f = lookup_format(texture_format);
	if (!f) { detexSetErrorMessage("detexhipModeHistogram: 0x%08X is not a block-compressed format of this library", texture_format); return false; }
	if (!context_ready()) return false;
	ThreadContext &c = t_ctx;
	DeviceScope scope(c.device);
	if (!scope.ok) return false;
	const size_t nbytes = n_blocks * detexGetCompressedBlockSize(texture_format);
	if (!reserve(&c.d_in, &c.in_cap, nbytes ? nbytes : 256)) return false;
	if (nbytes) HIP_TRY(hipMemcpyAsync(c.d_in, blocks, nbytes, hipMemcpyHostToDevice, c.stream), "hipMemcpyAsync(H2D)");
	uint32_t *d_hist = c.d_status;		// 16 words (the status allocation is 64 bytes)
	if (detexhipModeHistogramDevice(texture_format, c.d_in, n_blocks, d_hist, c.stream) != 0) return false;
	HIP_TRY(hipMemcpyAsync(histogram, d_hist, 64, hipMemcpyDeviceToHost, c.stream), "hipMemcpyAsync(D2H)");
	HIP_TRY(hipStreamSynchronize(c.stream), "hipStreamSynchronize");
	return true;
}

extern "C" bool detexDecompressTextureTiled(const detexTexture *texture, uint8_t *pixel_buffer, uint32_t pixel_format) {
	return decompress_texture(texture, pixel_buffer, pixel_format, true);
}

extern "C" bool detexDecompressTextureLinear(const detexTexture *texture, uint8_t *pixel_buffer, uint32_t pixel_format) {
	return decompress_texture(texture, pixel_buffer, pixel_format, false);
}

// The same two drivers under names that do not collide with the reference's: for a libdetex that forwards its own
// detexDecompressTextureLinear / Tiled here while both libraries are linked.
extern "C" bool detexhipHostDecompressTextureLinear(const detexTexture *texture, uint8_t *pixel_buffer, uint32_t pixel_format) {
	return decompress_texture(texture, pixel_buffer, pixel_format, false);
}
extern "C" bool detexhipHostDecompressTextureTiled(const detexTexture *texture, uint8_t *pixel_buffer, uint32_t pixel_format) {
	return decompress_texture(texture, pixel_buffer, pixel_format, true);
}

// ------------------------------------------------------------------------------------------------
// data symbols inlined helpers of the reference's detex.h refer to (detex.h:933,954,960,974):
// generated at compile time, value = clamp / truncating division (never used by this library)
// ------------------------------------------------------------------------------------------------
namespace {
template <int N> struct ByteTable { uint8_t v[N]; };
template <int N, int D> constexpr ByteTable<N> make_division_table() {
	ByteTable<N> t{};
	for (int i = 0; i < N; i++) t.v[i] = (uint8_t)(i / D);
	return t;
}
constexpr ByteTable<767> make_clamp_table() {
	ByteTable<767> t{};
	for (int i = 0; i < 767; i++) t.v[i] = (uint8_t)(i < 255 ? 0 : (i > 510 ? 255 : i - 255));
	return t;
}
}  // namespace
extern "C" {
__attribute__((visibility("default"))) extern const ByteTable<767> detex_clamp0to255_table_storage __asm__("detex_clamp0to255_table");
__attribute__((visibility("default"))) extern const ByteTable<768> detex_division_by_3_table_storage __asm__("detex_division_by_3_table");
__attribute__((visibility("default"))) extern const ByteTable<1792> detex_division_by_7_table_storage __asm__("detex_division_by_7_table");
__attribute__((visibility("default"))) extern const ByteTable<1280> detex_division_by_5_table_storage __asm__("detex_division_by_5_table");
const ByteTable<767> detex_clamp0to255_table_storage = make_clamp_table();
const ByteTable<768> detex_division_by_3_table_storage = make_division_table<768, 3>();
const ByteTable<1792> detex_division_by_7_table_storage = make_division_table<1792, 7>();
const ByteTable<1280> detex_division_by_5_table_storage = make_division_table<1280, 5>();
}
