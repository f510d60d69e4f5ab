// formats_bptc_float.hip -- kernels and launchers of BPTC_FLOAT / BPTC_SIGNED_FLOAT (BC6H, decompress-bptc-float.c): format
// indices 9-10 of the table the reference keeps in texture.c:27-48 -- and the half -> 8-bit table behind the FLOAT_RGBX16 -> 8-bit
// epilogues (kernels.h: kHalfToU8), which only this translation unit's kernels read.
#include <cmath>
#include <mutex>

#include "decode_bptc_float.h"
#include "launchers.h"

namespace detexhip {

// (resident workgroups per CU: BC6H gains 11 % on coherent content at five -- its fixture tiled, which is what encoder-made textures
// look like -- at the price of 2.6 % on uniform-random blocks, where the kernel sits at the board's power cap and needs every wave;
// signed BC6H has no fixture to show a gain and keeps all of them.  Block-major: four until round 6, uncapped since -- 99.9 -> 93.7 us with the
// blocks coming out of HBM and 85.9 -> 83.5 on a repeated input (profiles/r06/rotating_wg_sweep_tiled.txt).)
// (a function-local table: a namespace-scope const object would also be emitted into the device code object, where the launchers do not exist)
const FormatEntry *formats_bptc_float() {
	static const FormatEntry rows[2] = {
		FMT_RA(BPTC_FLOAT, DecBPTCFloat, kClassBPTCFloat, 5, 0), FMT(BPTC_SIGNED_FLOAT, DecBPTCSignedFloat, kClassBPTCFloat, 0, 0),
	};
	return rows;
}

// The FLOAT_RGBX16 -> 8-bit epilogues look each half up in kHalfToU8 (kernels.h).  Entry = the reference's
// FLOAT_RGBX16 -> RGBX16 -> RGBX8 path: f = half as float (exact), clamped to 0..1 (detex.h:941-948); u16 =
// lrintf(f * 65535.0f + 0.5f) with the multiply, the add and the conversion all rounding DOWN (half-float.c:304-312 sets
// FE_DOWNWARD); u8 = (u16 + 127) * 255 / 65535 (convert.c:299-313).  The float operations are reproduced in double
// (products and sums of these operands are exact there) and rounded down to float by hand, so the table does not depend
// on this translation unit's floating-point environment.  tests/test_oracle_pin.py compares all 65536 entries' effect
// with the compiled reference.
static float round_down_to_float(double v) {
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
static std::mutex g_half_table_mutex;
static bool g_half_table_ready[64];
// (uploaded on `stream` and waited for there -- not through hipMemcpyToSymbol, which goes through the NULL stream: in a process that
// has used no other stream of its own that is one more hardware queue to create, 13 of the 16 ms this call took in the one-shot client's
// API trace, profiles/r06/oneshot/)
hipError_t ensure_half_table(hipStream_t stream) {
	int dev = 0;
	hipError_t e = hipGetDevice(&dev);
	if (e != hipSuccess) return e;
	if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
	std::lock_guard<std::mutex> lock(g_half_table_mutex);
	if (g_half_table_ready[dev]) return hipSuccess;
	static uint8_t table[65536];
	static bool built = false;
	if (!built) { for (uint32_t h = 0; h < 65536u; h++) table[h] = half_to_u8_entry(h); built = true; }
	void *d_table = nullptr;
	if ((e = hipGetSymbolAddress(&d_table, HIP_SYMBOL(kHalfToU8))) != hipSuccess) return e;
	if ((e = hipMemcpyAsync(d_table, table, sizeof table, hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
	if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
	g_half_table_ready[dev] = true;
	return hipSuccess;
}

}  // namespace detexhip
