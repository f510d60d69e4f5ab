// measurement build only (make lib-ab): the product's BC7 table with the A/B linear launcher (ab_dispatch.h)
#include "decode_bptc.h"
#include "ab_dispatch.h"
#include "decode_bptc_r01.h"		// variants 3 / 4: the round-1 decoders
#include "kernels_sorted.h"		// variant 5: mode-sorted waves
namespace detexhip {
template <> struct AltDecoder<DecBPTC> { using type = r01::DecBPTCRegisterSelect; };
template <> struct AltDecoder2<DecBPTC> { using type = r01::DecBPTCLdsFields; };
}  // namespace detexhip
#include "formats_bptc.hip"
