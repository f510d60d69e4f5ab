// device_tier.cpp -- the device-pointer tier of libdetexhip (include/detexhip.h): format lookup, the target pixel formats of the
// block-decode path, the calling thread's settings (device, quirk mask, kernel variant) and the detexhip*Device entry points --
// argument validation, then one asynchronous launch through the format's row of launchers.  Host code only.
#include <cstdlib>

#include "host_internal.h"
#include "tune.h"

namespace detexhip {

// texture.c:55-70 dispatches by format >> 24 into an unchecked table; this lookup checks the index and the whole format word
const FormatEntry *lookup_format(uint32_t texture_format) {
	const uint32_t idx = texture_format >> 24;
	const FormatEntry *f = idx >= 1 && idx <= 8 ? formats_s3tc_rgtc() + (idx - 1) : idx >= 9 && idx <= 10 ? formats_bptc_float() + (idx - 9) :
		idx == 11 ? formats_bptc() : idx >= 12 && idx <= 19 ? formats_etc_eac() + (idx - 12) : nullptr;
	return f && f->texture_format == texture_format ? f : nullptr;
}

// Target pixel formats of the block-decode path and the epilogue that produces each (-1 = not offered): the native
// one, the RGBX8 <-> RGBA8 no-op edge (convert.c:768-769, 1087-1092), and -- converted inside the kernel with the exact
// result of the path detexConvertPixels takes (kernels.h) -- what the reference's callers request: BGRA8 / BGRX8
// (validate.c:204-209, detex-view.c:182), RGB8 (detex-convert.c:283-284) and RGBA8 / RGBX8, for every format the
// reference itself can convert (checked against the compiled reference: tools/make_goldens.py); FLOAT_BGRX16 for BC6H.
// Like the reference, the signed 16-bit formats have no path to BGRA8 and BPTC_SIGNED_FLOAT none to any 8-bit format.
enum : uint32_t { kPixelBGRA8 = 0x33C, kPixelBGRX8 = 0x328, kPixelRGB8 = 0x220, kPixelFloatBGRX16 = 0x2729 };
int epilogue_for(uint32_t texture_format, uint32_t pixel_format) {
	const uint32_t native = texture_format & DETEX_TEXTURE_FORMAT_PIXEL_FORMAT_MASK;
	if (pixel_format == native) return kEpiNone;
	const bool to_rgbx = pixel_format == DETEX_PIXEL_FORMAT_RGBA8 || pixel_format == DETEX_PIXEL_FORMAT_RGBX8;
	const bool to_bgrx = pixel_format == kPixelBGRA8 || pixel_format == kPixelBGRX8;
	if (native == DETEX_PIXEL_FORMAT_RGBA8 || native == DETEX_PIXEL_FORMAT_RGBX8)
		return to_rgbx ? kEpiNone : (to_bgrx ? kEpiSwapRB8 : (pixel_format == kPixelRGB8 ? kEpiPackRGB8 : -1));
	if (native == DETEX_PIXEL_FORMAT_FLOAT_RGBX16 && pixel_format == kPixelFloatBGRX16) return kEpiSwapRB16;
	const bool unsigned_small = native == DETEX_PIXEL_FORMAT_R8 || native == DETEX_PIXEL_FORMAT_RG8 || native == DETEX_PIXEL_FORMAT_R16 ||
		native == DETEX_PIXEL_FORMAT_RG16 || native == DETEX_PIXEL_FORMAT_FLOAT_RGBX16;
	const bool signed_small = native == DETEX_PIXEL_FORMAT_SIGNED_R16 || native == DETEX_PIXEL_FORMAT_SIGNED_RG16;
	if (unsigned_small || signed_small) {
		if (to_rgbx) return kEpiToRGBX8;
		if (to_bgrx) return (signed_small && pixel_format == kPixelBGRA8) ? -1 : kEpiToBGRX8;
		if (pixel_format == kPixelRGB8) return kEpiToRGB8;
	}
	return -1;
}
bool pixel_format_accepted(uint32_t texture_format, uint32_t pixel_format) { return epilogue_for(texture_format, pixel_format) >= 0; }
// a pixel format with the pixel size the epilogue `epi` writes for this texture format (for address arithmetic: bytes per block = 16 x its size)
static uint32_t epilogue_target_of(uint32_t texture_format, int epi) {
	switch (epi) {
	case kEpiPackRGB8: case kEpiToRGB8: return kPixelRGB8;
	case kEpiToRGBX8: case kEpiToBGRX8: return DETEX_PIXEL_FORMAT_RGBX8;
	default: return texture_format & DETEX_TEXTURE_FORMAT_PIXEL_FORMAT_MASK;		// native size (kEpiNone and the channel swaps)
	}
}

int prepared_epilogue(uint32_t texture_format, uint32_t pixel_format, hipStream_t stream) {
	const int epi = epilogue_for(texture_format, pixel_format);
	if ((texture_format & DETEX_TEXTURE_FORMAT_PIXEL_FORMAT_MASK) == DETEX_PIXEL_FORMAT_FLOAT_RGBX16 && epi >= kEpiToRGBX8) {
		hipError_t e = ensure_half_table(stream);
		if (e != hipSuccess) { detexSetErrorMessage("libdetexhip: half-float table upload failed: %s", hipGetErrorString(e)); return -2; }
	}
	return epi;
}

// The device tier launches on the caller's stream and keeps per-device state (the half-float table): a stream of another
// device than the current one would have the table uploaded to the wrong GPU and the kernel read zeros.  Refused instead.
bool stream_on_current_device(hipStream_t stream, const char *who) {
	if (stream == nullptr) return true;		// the null stream is the current device's
	hipDevice_t sdev = -1;
	int cur = -1;
	if (hipStreamGetDevice(stream, &sdev) != hipSuccess || hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); return true; }	// cannot tell: as before
	if ((int)sdev == cur) return true;
	detexSetErrorMessage("%s: the stream belongs to device %d but device %d is current (hipSetDevice to the stream's device before calling)", who, (int)sdev, cur);
	return false;
}

// ---- the calling thread's settings ---------------------------------------------------------------------------------------------
static thread_local ThreadSettings t_settings;
ThreadSettings &thread_settings() { return t_settings; }

// The product library has ONE kernel per format and layout (variant 0).  A measurement build (tools/ab: its own translation units, which
// include this library's headers and replace the format tables' launchers) raises the limit from a static initialiser.
int g_max_variant = 0;
int max_variant() { return g_max_variant; }
int current_variant() {
	ThreadSettings &s = t_settings;
	if (s.variant < 0) {
		const char *env = getenv("DETEXHIP_VARIANT");	// (always 0 in the product library: the limit above)
		s.variant = env ? atoi(env) : 0;
		if (s.variant < 0 || s.variant > max_variant()) s.variant = 0;
	}
	return s.variant;
}

// the reference's two BPTC quirks are reproduced unless switched off for the calling thread
// (detexhipSetQuirks, or DETEXHIP_QUIRKS in the environment when the thread first decodes); returns the decoders' spec flags
uint32_t current_spec_flags() {
	ThreadSettings &s = t_settings;
	if (s.quirks < 0) {
		const char *env = getenv("DETEXHIP_QUIRKS");
		s.quirks = env ? (int)(strtoul(env, nullptr, 0) & DETEXHIP_QUIRKS_REFERENCE) : (int)DETEXHIP_QUIRKS_REFERENCE;
	}
	return ((s.quirks & DETEXHIP_QUIRK_BC7_MODE6_PBIT) ? 0u : kFlagSpecBc7Mode6PBit) | ((s.quirks & DETEXHIP_QUIRK_BC6H_MODE12_BIT63) ? 0u : kFlagSpecBc6hMode12Bit63);
}

// Read-ahead of the blocks of textures beyond the Infinity Cache (linear_device_with): on unless switched off for the calling thread
// (detexhipSetReadAhead, or DETEXHIP_READ_AHEAD=0 in the environment when the thread first decodes)
// 0 = never, 1 = textures whose blocks exceed the Infinity Cache (default), 2 = every texture with at least 1 MiB of blocks (a caller
// who knows its blocks are NOT in that cache: freshly produced input, a stream of different textures)
int current_read_ahead() {
	ThreadSettings &s = t_settings;
	if (s.read_ahead < 0) { const char *env = getenv("DETEXHIP_READ_AHEAD"); s.read_ahead = env ? atoi(env) : 1; if (s.read_ahead < 0 || s.read_ahead > 2) s.read_ahead = 1; }
	return s.read_ahead;
}

int linear_device_with(uint32_t texture_format, const void *d_blocks, int width, int height, int width_in_blocks, int height_in_blocks,
		void *d_pixels, size_t pitch_bytes, uint32_t pixel_format, void *stream, uint32_t *d_status, uint32_t decode_flags, int variant, int read_ahead) {
	const FormatEntry *f = lookup_format(texture_format);
	if (!f) { detexSetErrorMessage("detexhipDecompressTextureLinearDevice: 0x%08X is not a block-compressed format of this library", texture_format); return 1; }
	if (!pixel_format_accepted(texture_format, pixel_format)) {
		detexSetErrorMessage("detexhipDecompressTextureLinearDevice: pixel format 0x%08X is outside the block-decode path for format 0x%08X", pixel_format, texture_format);
		return 1;
	}
	const size_t px = (size_t)detexGetPixelSize(pixel_format);
	const size_t palign = px == 3 ? 1 : (px < 4 ? px : 4);	// 24-bit pixels are byte-addressed, 64-bit ones dword-addressed
	if (width < 0 || height < 0 || width_in_blocks < 0 || height_in_blocks < 0 || pitch_bytes < (size_t)width * px ||
			(pitch_bytes % palign) != 0 || (reinterpret_cast<uintptr_t>(d_pixels) % palign) != 0 ||
			(uint64_t)width_in_blocks * (uint64_t)height_in_blocks > 0xFFFFFF00ull) {
		detexSetErrorMessage("detexhipDecompressTextureLinearDevice: bad geometry %dx%d (%dx%d blocks, pitch %zu)", width, height, width_in_blocks, height_in_blocks, pitch_bytes);
		return 1;
	}
	if (reinterpret_cast<uintptr_t>(d_blocks) % detexGetCompressedBlockSize(texture_format) != 0) {	// blocks are fetched with one 8/16-byte load each
		detexSetErrorMessage("detexhipDecompressTextureLinearDevice: d_blocks must be %d-byte aligned", (int)detexGetCompressedBlockSize(texture_format));
		return 1;
	}
	if (!stream_on_current_device(static_cast<hipStream_t>(stream), "detexhipDecompressTextureLinearDevice")) return 1;
	const int epi = prepared_epilogue(texture_format, pixel_format, static_cast<hipStream_t>(stream));
	if (epi == -2) return 1;
	Geometry g{ d_blocks, d_pixels, (uint32_t)width_in_blocks, (uint32_t)height_in_blocks, (uint32_t)width, (uint32_t)height,
		(uint64_t)pitch_bytes, d_status, static_cast<hipStream_t>(stream), variant, epi, decode_flags, f->resident };
	// (a texture whose blocks + pixels exceed the Infinity Cache may want another residency cap: FormatEntry::resident_beyond_cache)
	if (f->resident_beyond_cache >= 0 &&
			(size_t)width_in_blocks * (size_t)height_in_blocks * (detexGetCompressedBlockSize(texture_format) + 16u * px) > Tune::kInfinityCacheBytes)
		g.resident = f->resident_beyond_cache;
	// Blocks that cannot be resident in the Infinity Cache (more of them than it holds) come out of HBM in the middle of the write stream,
	// and HBM serves a read scattered among writes three times slower than a read from that cache (TCC_EA0_RDREQ_LEVEL / RDREQ: 2980 vs
	// 1000 cycles; the whole 32768^2 BC1 image 815 us where its four quarters, decoded alone, take 4 x 163: profiles/r06/footprint/).
	// Such textures go in bands of block rows -- a band is one contiguous range of blocks and of image rows, texture.c:115-141 -- each
	// band's blocks read into the cache by a read-only pass first: a read phase and a write phase per band, on the caller's stream -- for
	// the formats where two phases beat the mixed stream (FormatEntry::read_ahead_pays: BC6H, BC1 / BC1A; launchers.h has the table).
	const size_t bs = detexGetCompressedBlockSize(texture_format), row_bytes = (size_t)width_in_blocks * bs;
	const bool whole_grid = (size_t)width_in_blocks * 4u == (size_t)width && (size_t)height_in_blocks * 4u == (size_t)height;
	const size_t block_bytes_total = row_bytes * (size_t)height_in_blocks;
	if (whole_grid && row_bytes > 0 && row_bytes <= Tune::kReadAheadBandBytes &&
			((read_ahead == 1 && f->read_ahead_pays && block_bytes_total > Tune::kInfinityCacheBytes) || (read_ahead == 2 && block_bytes_total >= ((size_t)1 << 20)))) {
		const uint32_t band_rows = (uint32_t)(Tune::kReadAheadBandBytes / row_bytes);
		for (uint32_t r0 = 0; r0 < (uint32_t)height_in_blocks; r0 += band_rows) {
			const uint32_t rows = r0 + band_rows < (uint32_t)height_in_blocks ? band_rows : (uint32_t)height_in_blocks - r0;
			Geometry band = g;
			band.blocks = static_cast<const uint8_t *>(d_blocks) + (size_t)r0 * row_bytes;
			band.pixels = static_cast<uint8_t *>(d_pixels) + (size_t)r0 * 4u * pitch_bytes;
			band.hb = rows; band.height = rows * 4u;
			hipError_t e = launch_read_ahead(band.blocks, (size_t)rows * row_bytes, band.stream);
			if (e == hipSuccess) e = f->linear(band);
			if (e != hipSuccess) { detexSetErrorMessage("libdetexhip: kernel launch failed: %s", hipGetErrorString(e)); return 1; }
		}
		return 0;
	}
	hipError_t e = f->linear(g);
	if (e != hipSuccess) { detexSetErrorMessage("libdetexhip: kernel launch failed: %s", hipGetErrorString(e)); return 1; }
	return 0;
}

}  // namespace detexhip

using namespace detexhip;

// ---- library / thread management ---------------------------------------------------------------------------------------------
extern "C" int detexhipGetDeviceCount(void) {
	int n = 0;
	return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

extern "C" void detexhipReleaseThreadResources(void) { release_thread_context(); release_shard_slots(); }

extern "C" const char *detexhipVersion(void) { return "libdetexhip 0.4 (gfx950; detex v0.1.2 block-decode ABI; extension ABI 4)"; }
extern "C" int detexhipCheckAbi(int compiled_against) {
	if (compiled_against == DETEXHIP_ABI_VERSION) return 0;
	detexSetErrorMessage("libdetexhip: the caller was compiled against extension ABI %d, this library implements %d (struct layouts of detexhip.h differ)", compiled_against,
		DETEXHIP_ABI_VERSION);
	return 1;
}

extern "C" void detexhipSetQuirks(uint32_t quirks) { thread_settings().quirks = (int)(quirks & DETEXHIP_QUIRKS_REFERENCE); }
extern "C" uint32_t detexhipGetQuirks(void) { (void)current_spec_flags(); return (uint32_t)thread_settings().quirks; }

extern "C" int detexhipSetReadAhead(int mode) { const int before = current_read_ahead(); thread_settings().read_ahead = mode < 0 ? 0 : (mode > 2 ? 2 : mode); return before; }

extern "C" void detexhipSetKernelVariant(int variant) { thread_settings().variant = (variant >= 0 && variant <= max_variant()) ? variant : 0; }
extern "C" int detexhipGetKernelVariant(void) { return current_variant(); }

extern "C" const char *detexhipKernelName(uint32_t texture_format) {
	const FormatEntry *f = lookup_format(texture_format);
	if (!f) return nullptr;
	if (current_variant() == 1 && (texture_format >> 24) == 1) return "decode_linear_tile4x4";
	return f->kernel_name;
}

extern "C" uint8_t detexhipHalfFloatToUNorm8(uint16_t half_bits) { return half_to_u8_entry(half_bits); }

// ---- decode entry points ---------------------------------------------------------------------------------------------------------
extern "C" int detexhipDecompressTextureLinearDevice(uint32_t texture_format, const void *d_blocks, int width,
		int height, int width_in_blocks, int height_in_blocks, void *d_pixels, size_t pitch_bytes,
		uint32_t pixel_format, void *stream, uint32_t *d_status) {
	return linear_device_with(texture_format, d_blocks, width, height, width_in_blocks, height_in_blocks, d_pixels, pitch_bytes, pixel_format, stream, d_status,
		current_spec_flags(), current_variant(), current_read_ahead());
}

static int blocks_device(const char *who, uint32_t texture_format, const void *d_blocks, size_t n_blocks, uint32_t mode_mask,
		uint32_t flags, void *d_pixels, uint8_t *d_ok, uint32_t *d_status, void *stream, bool checked, int epi) {
	const FormatEntry *f = lookup_format(texture_format);
	if (!f) { detexSetErrorMessage("%s: 0x%08X is not a block-compressed format of this library", who, texture_format); return 1; }
	if (n_blocks > 0xFFFFFF00ull) { detexSetErrorMessage("%s: too many blocks", who); return 1; }
	if (reinterpret_cast<uintptr_t>(d_blocks) % detexGetCompressedBlockSize(texture_format) != 0 || reinterpret_cast<uintptr_t>(d_pixels) % 16u != 0) {
		detexSetErrorMessage("%s: d_blocks must be %d-byte aligned and d_pixels 16-byte aligned (block-major output is written with 16-byte vector stores)", who,
			(int)detexGetCompressedBlockSize(texture_format));
		return 1;
	}
	// the reference's flags occupy bits 0-2 (detex.h:397-411); the spec switches ride in bits 30-31
	BatchArgs a{ d_blocks, d_pixels, n_blocks, mode_mask, (flags & 0x3FFFFFFFu) | current_spec_flags(), d_ok, d_status, static_cast<hipStream_t>(stream), checked, epi, f->resident_blocks };
	// the same read-ahead as the linear layout (linear_device_with): a block-major stream is contiguous on both sides, so a band is simply a
	// run of blocks (whole 256-block tiles) and the pixels behind the run before it
	const size_t bs = detexGetCompressedBlockSize(texture_format), total = n_blocks * bs;
	const int read_ahead = current_read_ahead();
	if ((read_ahead == 1 && f->read_ahead_pays && total > Tune::kInfinityCacheBytes) || (read_ahead == 2 && total >= ((size_t)1 << 20))) {
		const size_t band_blocks = (Tune::kReadAheadBandBytes / bs) & ~(size_t)255, out_per_block = 16u * (size_t)detexGetPixelSize(epilogue_target_of(texture_format, epi));
		for (size_t b0 = 0; b0 < n_blocks; b0 += band_blocks) {
			BatchArgs band = a;
			band.n = b0 + band_blocks < n_blocks ? band_blocks : n_blocks - b0;
			band.blocks = static_cast<const uint8_t *>(d_blocks) + b0 * bs;
			band.pixels = static_cast<uint8_t *>(d_pixels) + b0 * out_per_block;
			band.ok = d_ok ? d_ok + b0 : nullptr;
			hipError_t e = launch_read_ahead(band.blocks, band.n * bs, band.stream);
			if (e == hipSuccess) e = f->blocks(band);
			if (e != hipSuccess) { detexSetErrorMessage("libdetexhip: kernel launch failed: %s", hipGetErrorString(e)); return 1; }
		}
		return 0;
	}
	hipError_t e = f->blocks(a);
	if (e != hipSuccess) { detexSetErrorMessage("libdetexhip: kernel launch failed: %s", hipGetErrorString(e)); return 1; }
	return 0;
}

extern "C" int detexhipDecompressTextureTiledDevice(uint32_t texture_format, const void *d_blocks, int width_in_blocks,
		int height_in_blocks, void *d_pixels, uint32_t pixel_format, void *stream, uint32_t *d_status) {
	if (lookup_format(texture_format) && !pixel_format_accepted(texture_format, pixel_format)) {
		detexSetErrorMessage("detexhipDecompressTextureTiledDevice: pixel format 0x%08X is outside the block-decode path for format 0x%08X", pixel_format, texture_format);
		return 1;
	}
	if (width_in_blocks < 0 || height_in_blocks < 0) { detexSetErrorMessage("detexhipDecompressTextureTiledDevice: bad geometry"); return 1; }
	if (!stream_on_current_device(static_cast<hipStream_t>(stream), "detexhipDecompressTextureTiledDevice")) return 1;
	const int epi = lookup_format(texture_format) ? prepared_epilogue(texture_format, pixel_format, static_cast<hipStream_t>(stream)) : kEpiNone;
	if (epi == -2) return 1;
	return blocks_device("detexhipDecompressTextureTiledDevice", texture_format, d_blocks,
		(size_t)width_in_blocks * (size_t)height_in_blocks, DETEX_MODE_MASK_ALL, 0, d_pixels, nullptr, d_status, stream, false, epi);
}

extern "C" int detexhipDecompressBlocksDevice(uint32_t texture_format, const void *d_blocks, size_t n_blocks, uint32_t mode_mask,
		uint32_t flags, void *d_pixels, uint8_t *d_ok, void *stream) {
	return blocks_device("detexhipDecompressBlocksDevice", texture_format, d_blocks, n_blocks, mode_mask, flags, d_pixels, d_ok,
		nullptr, stream, true, kEpiNone);
}

// 8f-3: up to 16 levels, one launch
extern "C" int detexhipDecompressLevelsLinearDevice(uint32_t texture_format, const detexhipLevel *levels, int n_levels,
		uint32_t pixel_format, void *stream, uint32_t *d_status) {
	const char *who = "detexhipDecompressLevelsLinearDevice";
	const FormatEntry *f = lookup_format(texture_format);
	if (!f) { detexSetErrorMessage("%s: 0x%08X is not a block-compressed format of this library", who, texture_format); return 1; }
	if (!stream_on_current_device(static_cast<hipStream_t>(stream), who)) return 1;
	const int epi = prepared_epilogue(texture_format, pixel_format, static_cast<hipStream_t>(stream));
	if (epi == -2) return 1;
	if (epi < 0) { detexSetErrorMessage("%s: pixel format 0x%08X is outside the block-decode path for format 0x%08X", who, pixel_format, texture_format); return 1; }
	if (n_levels < 0 || n_levels > kMaxLevels || (n_levels > 0 && !levels)) { detexSetErrorMessage("%s: 0..%d levels per call", who, kMaxLevels); return 1; }
	const size_t px = (size_t)detexGetPixelSize(pixel_format), palign = px == 3 ? 1 : (px < 4 ? px : 4);
	LevelsArgs a{};
	a.status = d_status; a.stream = static_cast<hipStream_t>(stream); a.epi = epi; a.decode_flags = current_spec_flags();
	a.table.n_levels = (uint32_t)n_levels;
	uint32_t wg = 0;
	for (int l = 0; l < n_levels; l++) {
		const detexhipLevel &s = levels[l];
		if (s.width < 0 || s.height < 0 || s.width_in_blocks < 0 || s.height_in_blocks < 0 || s.pitch_bytes < (size_t)s.width * px ||
				(s.pitch_bytes % palign) != 0 || (reinterpret_cast<uintptr_t>(s.d_pixels) % palign) != 0 ||
				(uint64_t)s.width_in_blocks * (uint64_t)s.height_in_blocks > 0x7FFFFF00ull) {
			detexSetErrorMessage("%s: bad geometry in level %d", who, l);
			return 1;
		}
		if (reinterpret_cast<uintptr_t>(s.d_blocks) % detexGetCompressedBlockSize(texture_format) != 0) {	// (one 8 / 16-byte load per block, as in the one-texture entry)
			detexSetErrorMessage("%s: d_blocks of level %d must be %d-byte aligned", who, l, (int)detexGetCompressedBlockSize(texture_format));
			return 1;
		}
		LevelDesc &d = a.table.level[l];
		d.blocks = s.d_blocks; d.pixels = static_cast<uint8_t *>(s.d_pixels); d.pitch = s.pitch_bytes;
		d.width_in_blocks = (uint32_t)s.width_in_blocks; d.n_blocks = (uint32_t)(s.width_in_blocks * s.height_in_blocks);
		d.width = (uint32_t)s.width; d.height = (uint32_t)s.height;
		a.table.wg_start[l] = wg;
		wg += (d.n_blocks + 255u) / 256u;
	}
	a.table.wg_start[n_levels] = wg;
	hipError_t e = f->levels(a);
	if (e != hipSuccess) { detexSetErrorMessage("libdetexhip: kernel launch failed: %s", hipGetErrorString(e)); return 1; }
	return 0;
}
