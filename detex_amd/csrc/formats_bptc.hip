// formats_bptc.hip -- kernels and launchers of BPTC (BC7, decompress-bptc.c): format index 11 of the table the reference keeps in
// texture.c:27-48.
#include "decode_bptc.h"
#include "launchers.h"

namespace detexhip {

// the kernels off the throughput path (clipped geometry, mip levels, the checked per-block batch) take the decoder without its
// wave-uniform per-record copies
template <> struct PlainDecoder<DecBPTCT<true>> { using type = DecBPTCPlain; };

const FormatEntry *formats_bptc() {
	static const FormatEntry rows[1] = { FMT(BPTC, DecBPTC, kClassBPTC, 0, 0) };
	return rows;
}

}  // namespace detexhip
