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

// ---- resident service of the host tier's smallest calls (kernels_resident.h, host_resident.cpp) --------------------------------------
// A launch + its completion cost a host-pointer caller 7.3 us for one block and 10.7 us for a 64x64 texture (the measured floor of a
// launch per call: profiles/r04/host_tier_latency.txt); a workgroup that is ALREADY RUNNING and polls a request line in pinned host
// memory answers in 4.2 / 6.4 us (tools/ubench/host_latency.hip: `resident`).  So the second small call in a row of one (format,
// target) pair starts such a kernel, later ones are posted to it, and it leaves by itself once no request has come for the idle time
// (detexhipSetResidentIdleMicroseconds; 100 us by default), after announcing that in `state`.
constexpr uint32_t kResidentWorkgroups = 4;					// textures of up to 4 x 256 blocks (128 x 128 pixels)
constexpr uint32_t kResidentMaxBlocks = 256u * kResidentWorkgroups;
constexpr uint32_t kResidentBlockBytes = kResidentMaxBlocks * 16u, kResidentPixelBytes = kResidentMaxBlocks * 16u * 8u;
enum : uint32_t { kResidentTexture = 0, kResidentBlock = 1, kResidentStop = 2, kResidentTagged = 3 };	// payload word 6
enum : uint32_t { kResidentRunning = 1, kResidentExiting = 2, kResidentExited = 3 };	// low two bits of `state`
// Pinned host memory, 64-byte lines with ONE writing side each.  A request is four 16-byte chunks {request number, three payload
// words}, each written by the host with one 16-byte store and read by the kernel with one 16-byte load: a request is taken when all
// four carry the same new number (no ordering between the four reads is assumed).  Payload words: 0-3 the block (kResidentBlock) or
// width, height, width_in_blocks, height_in_blocks; 4 mode_mask; 5 flags; 6 kind; 7 block-major output (texture.c:77-98); 8-11 unused.
// kResidentTagged = a texture of at most 256 blocks whose blocks the host has laid out as 16-byte chunks {eight block bytes, request
// number, 0}, too (one chunk per 8-byte block, two per 16-byte block): the leading workgroup reads its lanes' chunks WITH every poll
// of the request line, and a chunk that carries the request's number is that request's data whenever it was read -- so the second
// round trip across the link (request seen, then blocks fetched) is saved; chunks with another number are simply read again.
struct ResidentMail {
	uint32_t chunk[4][4];				// host -> device
	uint32_t done, pad0[15];			// device -> host: number of the last request completed
	uint32_t state, pad1[15];			// device -> host: instance << 2 | kResident{Running, Exiting, Exited}
	uint32_t status, pad2[15];			// host zeroes it with a request; the kernel raises it for a failed block
};
// `words` (device memory, zeroed once): [0] number of the request the leading workgroup handed to the others, [1] the instance that
// has left, [2..3] the workgroups' acknowledgement counter, keyed by instance and request (kernels_resident.h: resident_acknowledge),
// [4..15] that request's payload
struct ResidentArgs {
	ResidentMail *mail; const void *blocks; uint8_t *pixels; uint32_t *words;
	uint32_t start_seq, instance; uint64_t idle_ticks, max_ticks;	// ticks of the constant 100 MHz-class clock (wall_clock64)
	uint32_t speculative_polls;					// polls after a tagged request during which the leader's lanes read their chunks too
};

// ---- 8f-4: how a format's blocks are classified into modes (histogram.hip: block_mode) -------------------------------------------------
enum : int { kClassS3TC = 0, kClassS3TCat8, kClassETC1, kClassETC2, kClassETC2PT, kClassETC2at8, kClassBPTC, kClassBPTCFloat, kClassNone };

}  // namespace detexhip
