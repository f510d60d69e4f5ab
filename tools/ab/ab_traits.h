// ab/ab_traits.h -- primary templates of the rejected A/B kernels' per-decoder hooks (profiles/AB_RECORD.md).  Only compiled into the
// measurement build (make lib-ab); the product library contains none of this.  The tools/ab/formats_*_ab.hip translation unit that owns
// a decoder specialises the hooks that exist for it (variant_tile4x4.h: BC1; kernels_sorted.h, decode_bptc_r01.h: BC7; BC6H's
// switch-scatter decoder in formats_bptc_float_ab.hip) AFTER including ab_dispatch.h and before the product's FMT() rows.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace detexhip {

// variant 1: the north_star's 4x4-block wave tile
template <class Dec> struct Tile4x4 {
	static constexpr bool kAvailable = false;
	static hipError_t launch(const void *, uint8_t *, uint32_t, uint32_t, uint64_t, uint32_t *, hipStream_t) { return hipErrorNotSupported; }
};
// variant 5: mode-sorted waves
template <class Dec> struct ClassSorted {
	static constexpr bool kAvailable = false;
	template <int EPI> static hipError_t launch(const void *, uint8_t *, uint32_t, uint32_t, uint64_t, uint32_t *, hipStream_t) { return hipErrorNotSupported; }
};
// variants 3 / 4: alternative decoders of the same format
template <class Dec> struct AltDecoder { using type = Dec; };
template <class Dec> struct AltDecoder2 { using type = Dec; };

}  // namespace detexhip
