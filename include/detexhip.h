/*
 * include/detexhip.h -- device-pointer extension API of libdetexhip.
 *
 * The reference has no device boundary: every caller of its hot path
 * (detexDecompressTextureLinear, texture.c:105; callers validate.c:208, detex-view.c:182,
 * detex-convert.c:310) hands over HOST pointers.  libdetexhip keeps those entry points
 * (include/detex.h) and adds this second tier for callers whose blocks and pixels already
 * live in GPU memory -- the tier the roofline numbers are measured on.
 *
 * Plain C ABI: pointers, sizes, an opaque stream handle (a hipStream_t passed as void *;
 * NULL = the null stream).  No torch / C++ types cross this boundary.  All launches are
 * asynchronous on the given stream; nothing here synchronises.
 *
 * Return value: 0 on success (work enqueued), non-zero on a usage / HIP error, in which case
 * detexGetErrorMessage() (detex.h:806 convention, thread-local) describes it.
 */
#ifndef DETEXHIP_H
#define DETEXHIP_H

#include <stddef.h>
#include <stdbool.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DETEXHIP_API __attribute__((visibility("default")))

/* Library / device management.  detexhipSetDevice selects the GPU the HOST-POINTER tier of the calling thread
 * uses (default: DETEXHIP_DEVICE or 0); it does not change the thread's current HIP device -- the host tier
 * switches to its device for the duration of each call and restores the caller's.  The device tier below runs
 * wherever the caller's stream lives.  detexhipReleaseThreadResources frees the calling thread's staging
 * buffers, streams and events (they are re-created on the next host-tier call; a thread that decoded one
 * 32768^2 texture otherwise keeps 4.5 GiB of device memory for its lifetime). */
DETEXHIP_API int detexhipGetDeviceCount(void);
DETEXHIP_API int detexhipSetDevice(int device);
DETEXHIP_API void detexhipReleaseThreadResources(void);
/* Pixel buffers the kernels can write into directly.  The reference's callers malloc() the pixel_buffer they hand to
 * detexDecompressTextureLinear (validate.c:199); the GPU cannot reach such memory, so the host tier decodes into a buffer of its own and
 * copies (textures up to 1.25 MiB) or downloads (larger ones) into the caller's.  A pixel_buffer that lies inside memory returned by
 * detexhipAllocPixelBuffer is pinned and device-visible: linear textures with up to 8 MiB of pixels are then written by the kernel
 * straight into it -- same call, same result, no copy-out (256x256: 20 -> 15 us, 512x512: 50 -> 31 us, 1024x1024: 118 -> 108-111 us per
 * call from compiled C; the first two are the PCIe floor); larger ones are
 * downloaded into it at the link's rate.  One allocation may hold many images (any sub-range works), and the compressed blocks as well
 * (texture->data inside such memory is read by the kernel where it is, whatever its size: one copy less).  A sub-range that is not
 * aligned to the target pixel (4 bytes for 32- and 64-bit pixels) is decoded through the copying paths, like any other pointer.
 * Plain host memory otherwise: read and write it like malloc'ed memory, free it with detexhipFreePixelBuffer only.  NULL + error message
 * on failure.  Thread-safe: a decode holds the buffers it reads or writes directly until its kernel has completed, and
 * detexhipFreePixelBuffer WAITS for such decodes of other threads to end before it frees (if one has not ended after 10 s the buffer is
 * left allocated and the error message says so). */
DETEXHIP_API void *detexhipAllocPixelBuffer(size_t bytes);
DETEXHIP_API void detexhipFreePixelBuffer(void *pixel_buffer);
/* Test hook (tests/ only): the next `calls` host-pointer calls of the calling thread that launch a kernel return false right after the
 * launch, as if the runtime had failed there.  A call that fails after it launched leaves the thread's device state to be cleaned up by
 * the NEXT call (stream drained, status words zeroed): this is how the tests reach that path. */
DETEXHIP_API void detexhipTestFailAfterLaunch(int calls);
DETEXHIP_API const char *detexhipVersion(void);
/* The extension API's structs grow now and then (detexhipShard gained `peer_access` in 0.3 -> ABI 4): a client passes the
 * DETEXHIP_ABI_VERSION it was COMPILED with and gets 0 if this library lays the structs out the same way, non-zero (and an error
 * message) otherwise -- check once at start-up instead of having an array of shards read at the wrong stride.  The reference-tier
 * entry points of detex.h are not affected (their ABI is the reference's and does not change). */
#define DETEXHIP_ABI_VERSION 4
DETEXHIP_API int detexhipCheckAbi(int compiled_against);
/* The smallest host-pointer calls (one block: detexDecompressBlock*, detex.h:435-531 / texture.c:55-70; textures of up to
 * 1024 blocks, linear or block-major: texture.c:77-145) cost a kernel launch and its completion each -- 7 us / 11 us on the test box against the
 * reference's 3 us / 9 us on one host thread.  From the SECOND such call in a row of one (texture format, pixel format) pair on, the
 * calling thread's requests go to a kernel that stays resident and polls a request line in pinned host memory (4-5 us / 7 us per
 * call), and that leaves by itself once no request has come for `microseconds` (default: DETEXHIP_RESIDENT_US or 100).  While it
 * lingers a hipDeviceSynchronize() of the application waits that much longer.  0 = never leave a kernel behind (a launch per call).
 * Process-wide; applies to kernels started afterwards.  Returns the previous value. */
DETEXHIP_API int detexhipSetResidentIdleMicroseconds(int microseconds);
/* of the calling thread: requests answered by resident kernels so far, and resident kernels started (either pointer may be NULL) */
DETEXHIP_API void detexhipGetResidentStats(unsigned long long *requests, unsigned long long *instances);

/*
 * Device-resident counterpart of detexDecompressTextureLinear (texture.c:105-145).
 *   texture_format   DETEX_TEXTURE_FORMAT_* (detex.h:613-727)
 *   d_blocks         width_in_blocks*height_in_blocks blocks of 8/16 bytes, row-major
 *                    (texture.c:141), 8/16-byte aligned (checked: rc != 0 otherwise)
 *   width,height     image size in pixels; blocks are clipped to it (texture.c:116-136)
 *   d_pixels         row-major image, row r at d_pixels + r*pitch_bytes
 *   pitch_bytes      >= width*pixel_size; the reference's layout is exactly width*pixel_size.
 *                    Rows are written with 16-byte vector stores when width%4 == 0, d_pixels
 *                    and pitch_bytes are 16-byte aligned; any other geometry takes the clipped
 *                    per-pixel path (d_pixels, pitch_bytes aligned to the pixel size).
 *   pixel_format     native pixel format of texture_format; RGBA8/RGBX8 for either (the no-op
 *                    edge, convert.c:768-769); and, converted inside the kernel with the exact
 *                    result of the path detexConvertPixels takes (convert.c:885-1063): RGBA8,
 *                    RGBX8, BGRA8, BGRX8, RGB8 for every format the reference itself can convert
 *                    to them (all but BPTC_SIGNED_FLOAT; the signed 16-bit formats have no path
 *                    to BGRA8), FLOAT_BGRX16 for BPTC_FLOAT.  Anything else is refused (rc != 0,
 *                    nothing is written).  The first BPTC_FLOAT -> 8-bit call on a device uploads
 *                    a 64 KiB table synchronously (not capturable into a hipGraph; later calls are).
 *   d_status         optional device uint32_t: set to 1 by the kernel if any block was invalid
 *                    (those blocks are zero-filled, decoding continues: texture.c:125-128).
 *                    The caller zeroes it beforehand and reads it after synchronising;
 *                    result == (status == 0) is the reference's bool return value.
 */
DETEXHIP_API int detexhipDecompressTextureLinearDevice(uint32_t texture_format, const void *d_blocks,
	int width, int height, int width_in_blocks, int height_in_blocks,
	void *d_pixels, size_t pitch_bytes, uint32_t pixel_format, void *stream, uint32_t *d_status);

/*
 * One texture over several devices (the reference stores blocks row-major and writes a row-major
 * image, texture.c:115-141, so a band of block rows is ONE contiguous input range and ONE contiguous output
 * range -- the decode needs no exchange between devices).  Shard g of n decodes block rows
 * [g*hb/n, (g+1)*hb/n) (detexhipShardRows) on shards[g].device, from shards[g].d_blocks (that band's blocks,
 * resident on the device) or, when NULL, from the band's slice of host_blocks, uploaded by the call; into
 * shards[g].d_pixels (that band's rows, pitch_bytes apart; 0 = width*pixel_size).  One calling thread drives
 * all devices: per-shard streams, kernels launched back to back, HIP events for the per-device kernel time.
 * gather_device >= 0 additionally copies every band into the whole image d_gathered on that device with
 * hipMemcpyPeerAsync (direct xGMI transfers, one per source device, so all links of the root are busy -- where a peer mapping
 * exists: hipDeviceCanAccessPeer + hipDeviceEnablePeerAccess are tried per pair and the outcome is reported in shards[g].peer_access);
 * it is timed separately and is never part of the decode time.  The call returns after everything completed:
 * 0 = ran (the reference's bool result is "no shard has invalid_blocks"), non-zero = usage / HIP error.
 * *decode_wall_ms: host clock from the first launch until the last kernel finished (uploads excluded);
 * *gather_wall_ms: the same clock until the last peer copy finished.  Rows may be padded (pitch_bytes > width*pixel_size):
 * a band is then gathered row by row and the bytes between the rows are left alone on both sides.
 * Streams, events, status words and upload buffers are per calling thread and shard index, kept between calls
 * (detexhipReleaseThreadResources() frees them): no lock is taken, concurrent callers do not share anything, and peer
 * access is enabled once per process and device pair (a pair that could not be enabled is tried again by the next call).  On failure everything this call launched has completed
 * before it returns.
 */
typedef struct {
	int device;			/* in */
	const void *d_blocks;		/* in: the shard's blocks on `device`, or NULL (uploaded from host_blocks) */
	void *d_pixels;			/* in: the shard's band of the image on `device` */
	int row0, row1;			/* out: block rows [row0, row1) */
	float decode_ms;		/* out: kernel time on this device (HIP events) */
	int invalid_blocks;		/* out: 1 if a block of the band was invalid (zero-filled, texture.c:125-128) */
	int peer_access;		/* out: with a gather, 1 = `device` addresses gather_device's memory directly (the band travelled over the link
					 * between the two: xGMI on an MI355X node; also when they are the same device), 0 = no peer mapping (topology or
					 * runtime refused it): the runtime staged the copy -- gather_wall_ms is then NOT an xGMI figure; -1 = no gather */
} detexhipShard;
DETEXHIP_API int detexhipShardRows(int height_in_blocks, int n_shards, int shard, int *row0, int *row1);
DETEXHIP_API int detexhipDecompressTextureLinearMultiDevice(uint32_t texture_format, const void *host_blocks,
	int width, int height, int width_in_blocks, int height_in_blocks, size_t pitch_bytes, uint32_t pixel_format,
	detexhipShard *shards, int n_shards, int gather_device, void *d_gathered,
	float *decode_wall_ms, float *gather_wall_ms);

/*
 * Host image in, host image out, over several devices: detexDecompressTextureLinear's contract (host pointers, clipping,
 * invalid blocks zero-filled, only the pixels the block grid covers are written) with shard g of n_shards on devices[g] --
 * each shard uploads its band of blocks, decodes it and downloads its band of pixels over ITS OWN PCIe link (one worker
 * thread per shard; a device may be named more than once).  The single-device host tier is bounded by one link (~56 GB/s:
 * 8192^2 RGBA8 pixels take 4.8 ms to download against 0.04 ms of kernel).  *any_invalid = 1 if a block was invalid (the
 * reference's bool result is then false, and the same error text is left behind); returns non-zero on usage / HIP errors.
 * *wall_ms: the whole call.
 */
DETEXHIP_API int detexhipDecompressTextureLinearMultiDeviceHost(uint32_t texture_format, const void *host_blocks,
	int width, int height, int width_in_blocks, int height_in_blocks, void *host_pixels, size_t pitch_bytes,
	uint32_t pixel_format, const int *devices, int n_shards, int *any_invalid, float *wall_ms);

/* Device-resident counterpart of detexDecompressTextureTiled (texture.c:77-98): block i
 * occupies 16*pixel_size contiguous bytes of d_pixels (16-byte aligned); no clipping. */
DETEXHIP_API int detexhipDecompressTextureTiledDevice(uint32_t texture_format, const void *d_blocks,
	int width_in_blocks, int height_in_blocks, void *d_pixels, uint32_t pixel_format,
	void *stream, uint32_t *d_status);

/* Batched counterpart of the per-block functions detexDecompressBlock<FMT>
 * (detex.h:435-531): n_blocks independent blocks, honouring mode_mask and flags exactly as
 * the reference's leaf functions do.  Output is block-major, native pixel format; d_ok[i]
 * (uint8_t, optional) receives the leaf function's bool; failed blocks are zero-filled. */
DETEXHIP_API int detexhipDecompressBlocksDevice(uint32_t texture_format, const void *d_blocks, size_t n_blocks,
	uint32_t mode_mask, uint32_t flags, void *d_pixels, uint8_t *d_ok, void *stream);

/* The same on HOST pointers: the migration path of a client that loops over the leaf functions.  In the reference one
 * detexDecompressBlock<FMT> call (detex.h:435-531, e.g. decompress-bc.c:23-61) costs 0.03 us; here every call is a trip to the GPU
 * (5-8 us), and always will be -- so the loop
 *         for (i = 0; i < n; i++) ok &= detexDecompressBlockBC1(blocks + 8 * i, mode_mask, flags, pixels + 64 * i);
 * becomes ONE call
 *         ok = detexhipDecompressBlocks(DETEX_TEXTURE_FORMAT_BC1, blocks, n, mode_mask, flags, pixels, NULL);
 * blocks: n_blocks blocks of 8 / 16 bytes back to back (no alignment required); pixels: 16 pixels per block in the format's native pixel
 * format, block after block (what the leaf functions write); ok: optional, n_blocks bytes, ok[i] = the leaf function's bool for block i.
 * mode_mask and flags apply to every block, exactly as the leaf functions interpret them.  A failed block is zero-filled (the leaf
 * functions leave the caller's buffer in an unspecified state there).  Returns true if every block decoded, else false with the
 * reference's error text (the texture drivers' convention, texture.c:125-128,144); false without touching ok on usage / HIP errors.
 * Synchronous, per-thread state only, re-entrant like the rest of the host tier.  (INTEGRATION.md section 6 has the measured crossover.) */
DETEXHIP_API bool detexhipDecompressBlocks(uint32_t texture_format, const uint8_t *blocks, size_t n_blocks,
	uint32_t mode_mask, uint32_t flags, uint8_t *pixels, uint8_t *ok);

/* ---- mip-chain batching -------------------------------------------------------
 * Every level of a mip chain (what detexLoadKTXFileWithMipmaps returns as separate textures,
 * ktx.c:108-171) decoded by ONE kernel launch over a descriptor table; levels follow the same
 * rules as detexhipDecompressTextureLinearDevice (clipping, fast/clipped store path per level).
 * At most 16 levels per call (a full chain of a 32768^2 texture). */
typedef struct {
	const void *d_blocks;		/* width_in_blocks * height_in_blocks blocks, row-major; aligned to the block size (8 / 16 bytes), as for the one-texture entry */
	void *d_pixels;
	size_t pitch_bytes;
	int width, height, width_in_blocks, height_in_blocks;
} detexhipLevel;
DETEXHIP_API int detexhipDecompressLevelsLinearDevice(uint32_t texture_format, const detexhipLevel *levels,
	int n_levels, uint32_t pixel_format, void *stream, uint32_t *d_status);

/* Host-pointer form: n textures of one format (e.g. the array from detexLoadKTXFileWithMipmaps),
 * each decoded into pixel_buffers[i] exactly as detexDecompressTextureLinear would -- one staging
 * copy in, one launch, one copy out.  Returns false if any block of any level was invalid. */
#if !defined(__DETEX_H__) && !defined(DETEXHIP_COMPAT_DETEX_H)
#include "detex.h"
#endif
DETEXHIP_API bool detexhipDecompressTexturesLinear(const detexTexture *const *textures, int n_textures,
	uint8_t *const *pixel_buffers, uint32_t pixel_format);

/* detexDecompressTextureLinear / detexDecompressTextureTiled of this library under names that do not collide
 * with the reference's: what a libdetex that forwards its own drivers to the GPU would call when both
 * libraries are linked. */
DETEXHIP_API bool detexhipHostDecompressTextureLinear(const detexTexture *texture, uint8_t *pixel_buffer, uint32_t pixel_format);
DETEXHIP_API bool detexhipHostDecompressTextureTiled(const detexTexture *texture, uint8_t *pixel_buffer, uint32_t pixel_format);

/* ---- block-mode histogram -----------------------------------------------------
 * histogram[m] = number of blocks for which the reference's detexGetMode<FMT> returns m
 * (decompress-bc.c:63-69, decompress-etc.c:183-190,370-395,721-742, decompress-bptc.c:603-610,
 * decompress-bptc-float.c:647-658); bin 15 collects the reserved BPTC / BPTC_FLOAT codes (the
 * reference returns -1 there); formats without modes (RGTC, EAC R11/RG11) count into bin 0.
 * d_hist: 16 uint32_t, zeroed by the call. */
DETEXHIP_API int detexhipModeHistogramDevice(uint32_t texture_format, const void *d_blocks, size_t n_blocks,
	uint32_t *d_hist, void *stream);
/* The same count ADDED to d_hist (not zeroed): one histogram over several textures or the levels of a mip chain; a single
 * kernel launch, where the call above is a 64-byte memset plus the kernel.  Every bin must stay below 2^32 over the whole
 * accumulation (bins are added in pairs by 64-bit atomics: past the limit an even bin carries into its neighbour). */
DETEXHIP_API int detexhipModeHistogramAccumulateDevice(uint32_t texture_format, const void *d_blocks, size_t n_blocks,
	uint32_t *d_hist, void *stream);
DETEXHIP_API bool detexhipModeHistogram(uint32_t texture_format, const uint8_t *blocks, size_t n_blocks,
	uint32_t histogram[16]);

/* The 8-bit value the FLOAT_RGBX16 -> RGBA8/RGBX8/BGRA8/BGRX8/RGB8 epilogues write for one half-float component:
 * the reference's FLOAT_RGBX16 -> RGBX16 -> RGBX8 chain (half-float.c:304-312, convert.c:299-313).  Host function
 * (the kernels use a device table built from it); exposed so the table can be checked without a GPU. */
DETEXHIP_API uint8_t detexhipHalfFloatToUNorm8(uint16_t half_bits);

/* The reference's decoders differ from the BPTC specification in two places, and this library
 * reproduces both by default so that its output is the reference's, bit for bit:
 *   DETEXHIP_QUIRK_BC7_MODE6_PBIT      BC7 mode 6: the second endpoint's P-bit (block bit 64) reads 0 (decompress-bptc.c:142-146)
 *   DETEXHIP_QUIRK_BC6H_MODE12_BIT63   BC6H mode 12: block bit 63 (b0[11]) reads 0 in the reference build (decompress-bptc-float.c:462,
 *                                      an undefined shift in bits.h:29-31 as gcc >= -O2 compiles it)
 * detexhipSetQuirks(mask) chooses, for the calling thread, which of them stay on; a cleared bit gives the specification's
 * result for that case (what hardware decoders and other software decoders produce).  Default: DETEXHIP_QUIRKS_REFERENCE, or
 * the value of DETEXHIP_QUIRKS in the environment (read when the thread first decodes).  Applies to every entry point. */
enum { DETEXHIP_QUIRK_BC7_MODE6_PBIT = 1, DETEXHIP_QUIRK_BC6H_MODE12_BIT63 = 2, DETEXHIP_QUIRKS_REFERENCE = 3 };
DETEXHIP_API void detexhipSetQuirks(uint32_t quirks);
DETEXHIP_API uint32_t detexhipGetQuirks(void);

/* Read-ahead of the compressed blocks (detexhipDecompressTextureLinearDevice and everything built on it).  Blocks that come out of HBM in
 * the middle of the pixel write stream cost more than their bytes (HBM serves a read scattered among writes about three times slower);
 * blocks that sit in the GPU's 256 MiB memory-side cache (Infinity Cache) do not.  With read-ahead the texture is decoded in bands of block
 * rows, each band's blocks (at most 128 MiB) first read into that cache by a read-only pass -- several launches on the caller's stream
 * instead of one; same pixels, same status word.
 *   1 (default)  textures whose blocks alone exceed that cache and therefore cannot be resident in it, for the formats where two phases beat
 *                the mixed stream: BPTC_FLOAT (32768 x 32768, 1 GiB of blocks: 0.71 -> 0.80 of the HBM peak) and BC1 / BC1A (512 MiB: 0.77 ->
 *                0.80); the other formats' mixed stream already runs at what the memory allows, and they keep their single launch
 *   2            every texture with at least 1 MiB of blocks: for a caller who knows the blocks are NOT in the cache (freshly produced
 *                input, a stream of different textures); costs ~14 % where they are (a texture decoded again and again)
 *   0            never: always one launch
 * DETEXHIP_READ_AHEAD in the environment sets the initial mode.  Per calling thread.  Returns the previous mode. */
DETEXHIP_API int detexhipSetReadAhead(int mode);

/* Kernel-variant selection for A/B measurements (profiles/AB_RECORD.md).  The product library has ONE kernel per
 * format and layout (variant 0); the rejected alternatives exist only in the measurement build (make lib-ab:
 * the translation units of tools/ab, which include this library's sources; tools/ab/ab_dispatch.h lists them).  Unknown values fall back to 0.
 * Per calling thread.  Also settable with DETEXHIP_VARIANT. */
DETEXHIP_API void detexhipSetKernelVariant(int variant);
DETEXHIP_API int detexhipGetKernelVariant(void);

/* Name of the HIP kernel the linear-device entry would launch for this format with the
 * current variant (for matching rocprofv3 kernel-trace rows); NULL if unsupported. */
DETEXHIP_API const char *detexhipKernelName(uint32_t texture_format);

#ifdef __cplusplus
}
#endif
#endif
