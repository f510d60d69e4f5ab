// measurement build only (make lib-ab): the product's BC1-3 / RGTC table with the A/B linear launcher (ab_dispatch.h)
#include "decode_s3tc_rgtc.h"
#include "ab_dispatch.h"
#include "variant_tile4x4.h"		// variant 1: BC1 in 4x4-block wave tiles
namespace detexhip {
// variant 9: the row-split form of the two BC1 decoders -- `row_setup` is everything the block's sixteen texels share (the palette and the
// selector word, five dwords), `row_texels` picks texel row `row` (wave-uniform) from them
template <bool PUNCHTHROUGH> struct S3tcRowSplit {
	static constexpr bool kAvailable = true;
	static constexpr int kSharedDwords = 5;
	static DH void row_setup(uint2 blk, uint32_t (&s)[5]) {
		uint32_t p[4];
		s3tc_palette(blk.x, (blk.x & 0xFFFFu) > (blk.x >> 16), PUNCHTHROUGH ? 0u : 0xFF000000u, p);
		s[0] = p[0]; s[1] = p[1]; s[2] = p[2]; s[3] = p[3]; s[4] = blk.y;
	}
	static DH void row_texels(const uint32_t (&s)[5], uint32_t row, uint32_t (&o)[4]) {
		const uint32_t idx = s[4] >> (8u * row);
#pragma unroll
		for (int k = 0; k < 4; k++) o[k] = select4(bit_to_mask(idx, 2 * k), bit_to_mask(idx, 2 * k + 1), s[0], s[1], s[2], s[3]);
	}
};
template <> struct RowSplitOf<DecBC1> : S3tcRowSplit<false> {};
template <> struct RowSplitOf<DecBC1A> : S3tcRowSplit<true> {};
}  // namespace detexhip
#include "formats_s3tc_rgtc.hip"
