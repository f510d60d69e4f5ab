// path_types.h -- plain types and constants shared by the kernels (kernels.h, kernels_extra.h, the decoders) and the host-only
// translation units of the library (device_tier.cpp, host_tier.cpp, multi_device.cpp): no HIP device code in here.
#pragma once
#include <stdint.h>

namespace detexhip {

// ---- in-register pixel-format epilogues (kernels.h: Epilogue<>; reference: detexConvertPixels, convert.c) ----------------------------
enum : int {
	kEpiNone = 0,		// native pixel format (or the RGBX8 <-> RGBA8 no-op, convert.c:768-769)
	kEpiSwapRB8 = 1,	// RGBA8/RGBX8 -> BGRA8/BGRX8
	kEpiPackRGB8 = 2,	// RGBA8/RGBX8 -> RGB8: 4 pixels -> 3 dwords
	kEpiSwapRB16 = 3,	// FLOAT_RGBX16 -> FLOAT_BGRX16
	kEpiToRGBX8 = 4,	// 1/2-component and half-float natives -> RGBX8 / RGBA8 (4th byte 0xFF)
	kEpiToBGRX8 = 5,	//                                   ... -> BGRX8 / BGRA8
	kEpiToRGB8 = 6,		//                                   ... -> RGB8
};

// Spec-conformance switches, carried in the upper bits of the decoders' `flags` argument (the reference's own flags are
// bits 0-2, detex.h:397-411): the reference differs from the BPTC specification in two places, which
// the decoders reproduce unless these are set (detexhipSetQuirks clears the corresponding quirk).
enum : uint32_t {
	kFlagSpecBc7Mode6PBit = 1u << 30,	// BC7 mode 6: the second endpoint's P-bit is read from block bit 64 (the reference reads 0)
	kFlagSpecBc6hMode12Bit63 = 1u << 31,	// BC6H mode 12: block bit 63 (b0[11]) is used (the reference build drops it)
};

// ---- 8f-3: one launch over up to 16 mip levels (kernels_extra.h: decode_levels) ----------------------------------------------------
constexpr int kMaxLevels = 16;
struct LevelDesc {
	const void *blocks; uint8_t *pixels; uint64_t pitch;
	uint32_t width_in_blocks, n_blocks, width, height;
	uint32_t fast;			// 4-aligned geometry + vector-aligned rows: wave-wide row stores
	uint32_t pad;
};
struct LevelTable {
	uint32_t n_levels;
	uint32_t wg_start[kMaxLevels + 1];	// first workgroup of each level; [n_levels] = grid size
	LevelDesc level[kMaxLevels];
};

// ---- completion word of the host tier's small calls (host_tier.cpp: wait_for_ticket) ---------------------------------------------------
// A host-pointer caller of a small decode waits for ONE kernel.  Waiting through the runtime (hipStreamSynchronize: the command
// processor's end-of-kernel release, a signal, the runtime's wait) measured 12.5 / 14.3 us per call for a one-block / a 64x64 kernel
// on the test box; a word in pinned host memory that the kernel itself releases after its last store, polled by the caller, 9.0 /
// 10.9 (tools/ubench/host_latency.hip; profiles/r04/host_latency.txt).  `done` = device view of that word (nullptr: no completion
// signalling), `ticket` = the value to publish, `counter` = a zeroed device word the workgroups of a grid count themselves on (the
// last one to finish publishes, and zeroes it again for the next launch).
struct Completion { uint32_t *done; uint32_t *counter; uint32_t ticket; };

// ---- 8f-4: how a format's blocks are classified into modes (histogram.hip: block_mode) -------------------------------------------------
enum : int { kClassS3TC = 0, kClassS3TCat8, kClassETC1, kClassETC2, kClassETC2PT, kClassETC2at8, kClassBPTC, kClassBPTCFloat, kClassNone };

}  // namespace detexhip
