// measurement build only (make lib-ab): the product's BC6H table with the A/B linear launcher (ab_dispatch.h)
#include "decode_bptc_float.h"
#include "ab_dispatch.h"
namespace detexhip {
template <bool S> struct AltDecoder<DecBPTCFloatT<S, false>> { using type = DecBPTCFloatT<S, true>; };	// variant 3: field scatter as a per-mode switch
}  // namespace detexhip
#include "formats_bptc_float.hip"
