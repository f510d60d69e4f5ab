// ab/ab_dispatch.h -- launch-side selection of the rejected A/B kernels (profiles/AB_RECORD.md).
// Only compiled into the measurement build (make lib-ab: tools/ab/formats_*_ab.hip); the product library contains none of this.
//   1  4x4-block wave tiles staged through LDS, lane = (block, texel row)           [BC1 only]
//   2  default mapping with ordinary (cached) row stores
//   3  BPTC_FLOAT field scatter as a per-mode switch; BPTC round-1 decoder with register-select texel stage
//   4  BPTC round-1 default decoder (LDS block fields, per-texel index widths) -- the baseline decode_bptc.h replaced
//   5  BPTC with mode-sorted waves (workgroup counting sort by mode)
//   6  persistent grid, twice as many workgroups as are resident at once (kernels_persistent.h)      [32-bit pixels]
//   7  persistent grid, exactly the resident count
//   8  wave w of a 64-block workgroup decodes and stores texel row w: one store per lane, every block decoded four times (kernels_store_shape.h)   [32-bit pixels]
//   9  ... the cooperative form: wave 0 decodes palettes into LDS, wave w picks and stores row w                                                    [BC1 / BC1A]
//  10  one-wave workgroups (64 blocks, four stores per lane)                                                    [32-bit pixels, decoders without tables]
//  11  two tiles per workgroup, both blocks requested before the table copy                                                              [64-bit pixels]
// Included by the tools/ab/formats_*_ab.hip translation units AFTER the product's launchers.h; the per-decoder hooks are ab_traits.h's
// templates, specialised by the translation unit that owns the decoder.  The product sources know nothing of this directory: an A/B
// translation unit includes the product headers, re-points FMT()'s linear launcher at ab_linear<> below and then includes the product's
// formats_*.hip for its table rows.
#pragma once
#include "launchers.h"
#include "ab_traits.h"
#include "kernels_persistent.h"
#include "kernels_store_shape.h"

namespace detexhip {

// workgroups of `kernel` that are resident at once on the current device
template <class K> uint32_t resident_workgroups(K kernel) {
	int per_cu = 0, cus = 0, dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
			hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu <= 0 || cus <= 0)
		return 2048;
	return (uint32_t)per_cu * (uint32_t)cus;
}

// returns true if variant g.variant exists for <Dec, EPI> and was launched (*result = launch status)
template <class Dec, int EPI> bool ab_launch_linear(const Geometry &g, hipError_t *result) {
	const uint32_t n = g.wb * g.hb;
	const dim3 grid((n + 255u) / 256u), block(256);
	uint8_t *px = static_cast<uint8_t *>(g.pixels);
	if (EPI == kEpiNone && g.variant == 1 && Tile4x4<Dec>::kAvailable && (g.wb % 16u) == 0 && (g.hb % 4u) == 0) {
		*result = Tile4x4<Dec>::launch(g.blocks, px, g.wb, g.hb, g.pitch, g.status, g.stream);
		return true;
	}
	if constexpr (EPI == kEpiNone) {
		if (g.variant == 2) {
			hipLaunchKernelGGL((decode_linear<typename PlainDecoder<Dec>::type, kEpiNone, false>), grid, block, 0, g.stream, g.blocks, px, g.wb, n, g.pitch, g.status, 0u);
			*result = hipGetLastError();
			return true;
		}
	}
	if constexpr (EPI == kEpiNone && !std::is_same_v<typename AltDecoder<Dec>::type, Dec>) {
		if (g.variant == 3) {
			hipLaunchKernelGGL((decode_linear<typename AltDecoder<Dec>::type, kEpiNone, true>), grid, block, 0, g.stream, g.blocks, px, g.wb, n, g.pitch, g.status, 0u);
			*result = hipGetLastError();
			return true;
		}
	}
	if constexpr (EPI == kEpiNone && !std::is_same_v<typename AltDecoder2<Dec>::type, Dec>) {
		if (g.variant == 4) {
			hipLaunchKernelGGL((decode_linear<typename AltDecoder2<Dec>::type, kEpiNone, true>), grid, block, 0, g.stream, g.blocks, px, g.wb, n, g.pitch, g.status, 0u);
			*result = hipGetLastError();
			return true;
		}
	}
	if constexpr (ClassSorted<Dec>::kAvailable && EpilogueOf<Dec, EPI>::kRowDwords == 4) {
		if (g.variant == 5) {
			*result = ClassSorted<Dec>::template launch<EPI>(g.blocks, px, g.wb, n, g.pitch, g.status, g.stream);
			return true;
		}
	}
	if constexpr (EpilogueOf<Dec, EPI>::kRowDwords != 8) {
		if (g.variant == 6 || g.variant == 7) {
			auto kernel = decode_linear_persistent<Dec, EPI>;
			const uint32_t tiles = (n + 255u) / 256u, want = (g.variant == 6 ? 2u : 1u) * resident_workgroups(kernel);
			hipLaunchKernelGGL(kernel, dim3(tiles < want ? tiles : want), block, 0, g.stream, g.blocks, px, g.wb, n, g.pitch, g.status);
			*result = hipGetLastError();
			return true;
		}
	}
	if constexpr (EpilogueOf<Dec, EPI>::kRowDwords == 4) {
		if (g.variant == 8) {
			hipLaunchKernelGGL((decode_linear_rowwave<Dec, EPI>), dim3((n + 63u) / 64u), block, 0, g.stream, g.blocks, px, g.wb, n, g.pitch, g.status, g.decode_flags);
			*result = hipGetLastError();
			return true;
		}
		if constexpr (RowSplitOf<Dec>::kAvailable && EPI == kEpiNone) {
			if (g.variant == 9) {
				hipLaunchKernelGGL((decode_linear_rowsplit<Dec>), dim3((n + 63u) / 64u), block, 0, g.stream, g.blocks, px, g.wb, n, g.pitch);
				*result = hipGetLastError();
				return true;
			}
		}
		if constexpr (!HasTables<Dec>::value && EPI == kEpiNone) {
			if (g.variant == 10) {
				hipLaunchKernelGGL((decode_linear_onewave<Dec, EPI>), dim3((n + 63u) / 64u), dim3(64), 0, g.stream, g.blocks, px, g.wb, n, g.pitch, g.status, g.decode_flags);
				*result = hipGetLastError();
				return true;
			}
		}
	}
	if constexpr (EpilogueOf<Dec, EPI>::kRowDwords == 8) {
		if (g.variant == 11) {
			const uint32_t tiles = (n + 255u) / 256u;
			hipLaunchKernelGGL((decode_linear_wide_tiles<Dec, EPI, 2>), dim3((tiles + 1u) / 2u), block, 0, g.stream, g.blocks, px, g.wb, n, g.pitch, g.status, g.decode_flags);
			*result = hipGetLastError();
			return true;
		}
	}
	return false;
}

// the format table's linear launcher in the measurement build: variant 0, geometries the variants do not cover and variants a decoder
// does not have go to the product's launcher
template <class Dec> hipError_t ab_linear(const Geometry &g) {
	if (g.variant != 0 && g.wb * g.hb != 0) {
		hipError_t result = hipSuccess;
		bool launched = false;
		(void)with_epilogue<Dec>(g.epi, [&](auto epi) {
			constexpr int EPI = decltype(epi)::value;
			bool sector_aligned = false;
			constexpr bool kStagedWhenUnaligned = EpilogueOf<Dec, EPI>::kRowDwords >= 3;
			if (fast_geometry<Dec, EPI>(g, &sector_aligned) && (sector_aligned || !kStagedWhenUnaligned)) launched = ab_launch_linear<Dec, EPI>(g, &result);
			return hipSuccess;
		});
		if (launched) return result;
	}
	return launch_linear<Dec>(g);
}
static const int g_raise_variant_limit = (g_max_variant = 11);

}  // namespace detexhip

#undef FMT_ROW
#define FMT_ROW(NAME, DEC, CLS, RESIDENT, RESIDENT_BLOCKS, RESIDENT_LARGE, READ_AHEAD) { #NAME, DETEX_TEXTURE_FORMAT_##NAME, &ab_linear<DEC>, &launch_blocks<DEC>, &launch_single<DEC>, \
	&launch_levels<DEC>, &launch_resident<DEC>, CLS, "decode_linear<detexhip::" #DEC, RESIDENT, RESIDENT_BLOCKS, RESIDENT_LARGE, READ_AHEAD }
