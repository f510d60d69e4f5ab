// host_internal.h -- what the translation units of libdetexhip share on the host side (never installed; the public
// interface is include/detex.h + include/detexhip.h).
//
//   errors.cpp               error convention (misc.c:73-94), the LUT data symbols the reference header's inline helpers name
//   formats_s3tc_rgtc.hip    kernels + launchers of BC1/BC1A/BC2/BC3, RGTC1/2 +- signed     (decode_s3tc_rgtc.h)
//   formats_etc_eac.hip      ... ETC1, ETC2, punchthrough, ETC2_EAC, EAC R11/RG11 +- signed  (decode_etc_eac.h)
//   formats_bptc.hip         ... BPTC (BC7)                                                  (decode_bptc.h)
//   formats_bptc_float.hip   ... BPTC_FLOAT / BPTC_SIGNED_FLOAT (BC6H), the half -> 8-bit table of their 8-bit epilogues
//   histogram.hip            8f-4: mode histogram kernels and their entry points
//   device_tier.cpp          format lookup, target pixel formats, per-thread settings, the detexhip*Device entry points
//   host_tier.cpp            the reference's own entry points on host pointers (texture.c:55-145, the 19 leaf decoders)
//   host_resident.cpp        the resident service kernel behind the smallest of those calls (host side of kernels_resident.h)
//   multi_device.cpp         one texture over several devices
//   ktx_loader.cpp           8f-1
// Only the .hip files contain device code; each instantiates the kernels of its formats and exports one row of launchers per
// format (FormatEntry), which is all the host-only files know about the kernels.
#pragma once
#include <hip/hip_runtime_api.h>

#include <pthread.h>
#include <signal.h>
#include <stddef.h>
#include <stdint.h>
#include <utility>

#ifndef DETEXHIP_BUILDING_LIBRARY
#define DETEXHIP_BUILDING_LIBRARY 1
#endif
#include "../../include/detex.h"
#include "../../include/detexhip.h"
#include "path_types.h"

#define HIP_TRY(expr, what)                                                                      \
	do {                                                                                         \
		hipError_t e_ = (expr);                                                                  \
		if (e_ != hipSuccess) {                                                                  \
			detexSetErrorMessage("libdetexhip: %s failed: %s", what, hipGetErrorString(e_));     \
			return false;                                                                        \
		}                                                                                        \
	} while (0)

namespace detexhip {

// A helper thread of the library's own (the duplex upload of host_tier.cpp, the per-shard workers of multi_device.cpp): started with every
// signal blocked -- the host application's handlers run on the application's threads, not on one it does not know about -- and a refused
// thread reported instead of thrown through the C ABI.
template <class Thread, class F> inline bool start_helper_thread(Thread &t, F &&f) {
	sigset_t all, old;
	sigfillset(&all);
	const bool masked = pthread_sigmask(SIG_SETMASK, &all, &old) == 0;
	bool started = true;
	try { t = Thread(std::forward<F>(f)); } catch (...) { started = false; }
	if (masked) (void)pthread_sigmask(SIG_SETMASK, &old, nullptr);
	return started;
}

// ---- what a launcher is handed --------------------------------------------------------------------------------------------
struct Geometry {
	const void *blocks; void *pixels; uint32_t wb, hb, width, height; uint64_t pitch;
	uint32_t *status; hipStream_t stream; int variant; int epi; uint32_t decode_flags;
	int resident;		// workgroups per CU the linear kernel of this format runs best with (FormatEntry::resident; 0 = no cap)
};
struct BatchArgs {
	const void *blocks; void *pixels; size_t n; uint32_t mode_mask, flags; uint8_t *ok; uint32_t *status;
	hipStream_t stream; bool checked; int epi; int resident;
	Completion completion;	// {done != nullptr}: the small-batch kernel of the host tier (kernels_extra.h: decode_blocks_direct), which publishes it
};
// one block handed over as a kernel argument (kernels_extra.h: decode_single)
struct SingleArgs { const uint8_t *bitstring; uint32_t mode_mask, flags; uint32_t *pixels; uint8_t *ok; hipStream_t stream; int epi; uint32_t *done; uint32_t ticket; };
// 8f-3: all levels of a mip chain in one launch (kernels_extra.h: decode_levels)
struct LevelsArgs { LevelTable table; uint32_t *status; hipStream_t stream; int epi; uint32_t decode_flags; Completion completion; };
// the resident service kernel of the host tier's smallest calls (kernels_resident.h)
struct ResidentLaunch { ResidentArgs args; hipStream_t stream; int epi; };

// One row per block format (texture.c:27-48 is the reference's table of decompress functions): the launchers of its kernels.
struct FormatEntry {
	const char *name;
	uint32_t texture_format;
	hipError_t (*linear)(const Geometry &);
	hipError_t (*blocks)(const BatchArgs &);
	hipError_t (*single)(const SingleArgs &);
	hipError_t (*levels)(LevelsArgs &);
	hipError_t (*service)(const ResidentLaunch &);	// the resident service kernel (kernels_resident.h)
	int histogram_class;		// kClass... (histogram.hip)
	const char *kernel_name;
	int resident;			// resident workgroups per CU of the linear kernels (launchers.h: occupancy_cap_lds); 0 = whatever fits
	int resident_blocks;		// the same for the block-major texture driver
	int resident_beyond_cache;	// ... for the linear kernels when blocks + pixels exceed the Infinity Cache (Tune::kInfinityCacheBytes); -1 = the same as `resident`
	bool read_ahead_pays;		// textures whose blocks exceed the Infinity Cache go in bands behind a read-ahead pass by default (device_tier.cpp; launchers.h: FMT_RA)
};
// format index (texture_format >> 24, detex.h:913-915): 1-8, 9-10, 11, 12-19
const FormatEntry *formats_s3tc_rgtc(), *formats_bptc_float(), *formats_bptc(), *formats_etc_eac();	// 8, 2, 1, 8 rows (formats_*.hip)
const FormatEntry *lookup_format(uint32_t texture_format);		// nullptr: not a block format of this library (bounds-checked, SURVEY A-11)

// ---- target pixel formats (device_tier.cpp) ----------------------------------------------------------------------------------
int epilogue_for(uint32_t texture_format, uint32_t pixel_format);	// kEpi..., -1 = not offered
bool pixel_format_accepted(uint32_t texture_format, uint32_t pixel_format);
// epilogue for an accepted pair with its device table in place on the CURRENT device (-2 + error message if the upload failed); the
// first use on a device uploads the table on `stream` (the one the caller is about to launch on) and waits for it
int prepared_epilogue(uint32_t texture_format, uint32_t pixel_format, hipStream_t stream);
hipError_t ensure_half_table(hipStream_t stream);			// formats_bptc_float.hip
uint8_t half_to_u8_entry(uint32_t half_bits);				// formats_bptc_float.hip (host function)
bool stream_on_current_device(hipStream_t stream, const char *who);

// ---- per-thread settings (device_tier.cpp): detexhipSetDevice / SetQuirks / SetKernelVariant ------------------------------------
struct ThreadSettings { int device = -1, variant = -1, quirks = -1, read_ahead = -1; };	// -1 = not read yet (environment / default)
ThreadSettings &thread_settings();
int current_variant();
uint32_t current_spec_flags();						// the decoders' kFlagSpec... bits for the calling thread's quirk mask
int max_variant();							// 0 in the product library
extern int g_max_variant;					// (raised by the measurement build of tools/ab)

// detexhipDecompressTextureLinearDevice with the quirk flags, the kernel variant and the read-ahead mode given explicitly instead of read
// from the calling thread's settings: for worker threads that decode on behalf of a caller (multi_device.cpp)
int linear_device_with(uint32_t texture_format, const void *d_blocks, int width, int height, int width_in_blocks, int height_in_blocks,
	void *d_pixels, size_t pitch_bytes, uint32_t pixel_format, void *stream, uint32_t *d_status, uint32_t decode_flags, int variant, int read_ahead);
int current_read_ahead();						// 0 / 1 / 2 (detexhipSetReadAhead) of the calling thread

// ---- 8f-4 (histogram.hip) ----------------------------------------------------------------------------------------------------------
hipError_t launch_mode_histogram(int histogram_class, int block_dwords, const void *blocks, size_t n, uint32_t *hist, hipStream_t stream, bool zero_first);

// ---- read-ahead of compressed blocks into the Infinity Cache (histogram.hip; used by device_tier.cpp for inputs beyond that cache) ------------
hipError_t launch_read_ahead(const void *p, size_t bytes, hipStream_t stream);

// ---- resident service of the host tier's smallest calls (host_resident.cpp; protocol: path_types.h ResidentMail) ---------------------
// One per host-tier thread context.  All of it runs with the context's device current.
struct ResidentService {
	bool ready = false, launched = false, broken = false;
	const FormatEntry *f = nullptr;		// what the running (or last) instance decodes
	int epi = 0;
	uint32_t instance = 0, seq = 0;
	hipStream_t stream = nullptr;
	uint8_t *h_buf = nullptr, *d_buf = nullptr;	// [ResidentMail][blocks][pixels], pinned
	uint32_t *d_words = nullptr;
	uint64_t ticks_per_us = 100;
	uint64_t served = 0, started = 0;	// requests answered by resident kernels / instances started (detexhipGetResidentStats)
	const FormatEntry *prev_f = nullptr;	// the previous small call's pair: the service starts with the second call in a row of one pair
	int prev_epi = -1;
	// true if this small call should go through the service (enabled, not broken, the pair repeats, buffers in place); records the pair either way
	bool wanted(const FormatEntry *fmt, int epilogue);
	// Order of a call: wanted() -> begin() (stops an instance of another pair; gives the request its number) -> blocks into blocks_host()
	// or tagged with pack_tagged() -> serve().
	uint32_t begin(const FormatEntry *fmt, int epilogue);	// 0: error message set
	uint8_t *blocks_host();			// where the caller puts the blocks of a kResidentTexture request (kResidentBlockBytes)
	void pack_tagged(const void *blocks, size_t bytes, uint32_t number);	// the blocks of a kResidentTagged request (at most 256 blocks)
	const uint8_t *pixels_host() const;	// where the pixels of the last request are (kResidentPixelBytes)
	// posts the request `number` (payload words 0-11 of ResidentMail) and waits for it; *failed = some block was invalid.  false: error message set
	bool serve(const uint32_t payload[12], uint32_t number, bool *failed);
	void release();
private:
	uint32_t before = 0;			// the number before the one begin() handed out (what a new instance starts from)
	bool prepare(int device);
	bool launch(uint32_t start_seq);
	void post(const uint32_t payload[12], uint32_t number);
	bool stop(bool quiet = false);	// quiet: a failure sets no error message (the caller's call goes on without the service)
};
int resident_idle_microseconds();	// detexhipSetResidentIdleMicroseconds, else DETEXHIP_RESIDENT_US, else 100; 0 = no resident kernels

// ---- host tier <-> multi-device (host_tier.cpp, multi_device.cpp) --------------------------------------------------------------------
void release_thread_context();
void release_shard_slots();

}  // namespace detexhip
