// tests/host_san/api_san_main.cpp -- TEST-ONLY: the host side of libdetexhip (argument validation, error convention, the
// conversion-table builder) compiled with AddressSanitizer + UndefinedBehaviorSanitizer on the host code (flags: tests/host_san/san.mk;
// container only -- the GPU box runs the uninstrumented build, `make host-plain`) and called with hostile arguments.  Every call below must be REFUSED with an error
// message and must not touch memory it was not given; a sanitizer report aborts with a non-zero exit code.  Runs without a GPU
// (calls that pass validation then fail with "no usable HIP device"), and with one.  Not part of the product.
#include "../../include/detex.h"
#include "../../include/detexhip.h"
#include <cstdio>
#include <cstring>
#include <ctime>

#include <vector>

static int g_failures = 0;
#define REFUSED(expr)                                                                                   \
	do {                                                                                                \
		detexSetErrorMessage("(none)");                                                                 \
		const bool accepted = (expr);                                                                   \
		const char *m = detexGetErrorMessage();                                                         \
		if (accepted || !m || !strcmp(m, "(none)")) { printf("NOT REFUSED: %s (message: %s)\n", #expr, m ? m : "NULL"); g_failures++; } \
	} while (0)

int main() {
	std::vector<uint8_t> blocks(4096, 0x5A), pixels(1 << 16, 0);
	uint8_t *in = blocks.data(), *out = pixels.data();
	const uint32_t BC1 = DETEX_TEXTURE_FORMAT_BC1, RGBA8 = DETEX_PIXEL_FORMAT_RGBA8;
	// formats and pixel formats outside the path
	REFUSED(detexDecompressBlock(in, 0x00000320u, DETEX_MODE_MASK_ALL, 0, out, RGBA8));
	REFUSED(detexDecompressBlock(in, 0xFF000320u, DETEX_MODE_MASK_ALL, 0, out, RGBA8));
	REFUSED(detexDecompressBlock(in, 0x14000320u, DETEX_MODE_MASK_ALL, 0, out, RGBA8));
	REFUSED(detexDecompressBlock(in, BC1, DETEX_MODE_MASK_ALL, 0, out, 0x2721u));
	REFUSED(detexDecompressBlock(in, DETEX_TEXTURE_FORMAT_BPTC_SIGNED_FLOAT, DETEX_MODE_MASK_ALL, 0, out, RGBA8));
	// texture drivers: negative and inconsistent geometry, unknown formats (image zeroed, false)
	detexTexture t = { BC1, in, -4, 4, 1, 1 };
	REFUSED(detexDecompressTextureLinear(&t, out, RGBA8));
	t = detexTexture{ BC1, in, 4, 4, -1, 1 };
	REFUSED(detexDecompressTextureTiled(&t, out, RGBA8));
	t = detexTexture{ 0x15000320u, in, 8, 8, 2, 2 };
	memset(out, 0xEE, 256);
	REFUSED(detexDecompressTextureLinear(&t, out, RGBA8));
	for (int k = 0; k < 8 * 8 * 4; k++) if (out[k] != 0) { printf("unknown format: image not zeroed\n"); g_failures++; break; }
	if (out[8 * 8 * 4] != 0) { printf("unknown format: wrote past the image\n"); g_failures++; }
	t = detexTexture{ DETEX_PIXEL_FORMAT_RGBA8, in, 4, 4, 1, 1 };		// uncompressed: only the identity edge is on the path
	REFUSED(detexDecompressTextureLinear(&t, out, 0x0101u));
	REFUSED(detexDecompressTextureTiled(&t, out, RGBA8));
	if (!detexDecompressTextureLinear(&t, out, RGBA8) || memcmp(out, in, 64) != 0) { printf("identity copy failed\n"); g_failures++; }
	// device tier
	REFUSED(detexhipDecompressTextureLinearDevice(0x99000000u, in, 4, 4, 1, 1, out, 16, RGBA8, nullptr, nullptr) == 0);
	REFUSED(detexhipDecompressTextureLinearDevice(BC1, in, 4, 4, 1, 1, out, 16, 0x1234u, nullptr, nullptr) == 0);
	REFUSED(detexhipDecompressTextureLinearDevice(BC1, in, -4, 4, 1, 1, out, 16, RGBA8, nullptr, nullptr) == 0);
	REFUSED(detexhipDecompressTextureLinearDevice(BC1, in, 4, 4, 1, 1, out, 15, RGBA8, nullptr, nullptr) == 0);
	REFUSED(detexhipDecompressTextureLinearDevice(BC1, in, 8, 4, 2, 1, out, 16, RGBA8, nullptr, nullptr) == 0);
	REFUSED(detexhipDecompressTextureLinearDevice(BC1, in + 4, 4, 4, 1, 1, out, 16, RGBA8, nullptr, nullptr) == 0);
	REFUSED(detexhipDecompressTextureLinearDevice(BC1, in, 4, 4, 1, 1, out + 2, 16, RGBA8, nullptr, nullptr) == 0);
	REFUSED(detexhipDecompressTextureLinearDevice(BC1, in, 4, 4, 0x7FFFFFFF, 0x7FFFFFFF, out, 16, RGBA8, nullptr, nullptr) == 0);
	REFUSED(detexhipDecompressTextureTiledDevice(BC1, in, -1, 1, out, RGBA8, nullptr, nullptr) == 0);
	REFUSED(detexhipDecompressTextureTiledDevice(BC1, in, 1, 1, out + 4, RGBA8, nullptr, nullptr) == 0);
	REFUSED(detexhipDecompressTextureTiledDevice(BC1, in, 1, 1, out, 0x4444u, nullptr, nullptr) == 0);
	REFUSED(detexhipDecompressBlocksDevice(0, in, 1, DETEX_MODE_MASK_ALL, 0, out, nullptr, nullptr) == 0);
	REFUSED(detexhipDecompressBlocksDevice(BC1, in + 1, 1, DETEX_MODE_MASK_ALL, 0, out, nullptr, nullptr) == 0);
	REFUSED(detexhipDecompressBlocksDevice(BC1, in, (size_t)1 << 40, DETEX_MODE_MASK_ALL, 0, out, nullptr, nullptr) == 0);
	// the batched host-pointer block entry and the ABI check
	REFUSED(detexhipDecompressBlocks(0x14000320u, in, 4, DETEX_MODE_MASK_ALL, 0, out, nullptr));
	REFUSED(detexhipDecompressBlocks(BC1, nullptr, 4, DETEX_MODE_MASK_ALL, 0, out, nullptr));
	REFUSED(detexhipDecompressBlocks(BC1, in, 4, DETEX_MODE_MASK_ALL, 0, nullptr, nullptr));
	REFUSED(detexhipDecompressBlocks(BC1, in, (size_t)1 << 40, DETEX_MODE_MASK_ALL, 0, out, nullptr));
	if (!detexhipDecompressBlocks(BC1, nullptr, 0, DETEX_MODE_MASK_ALL, 0, nullptr, nullptr)) { printf("zero blocks must succeed\n"); g_failures++; }
	{ int not_ours = 0; detexSetErrorMessage("(none)"); detexhipFreePixelBuffer(&not_ours);		// a foreign pointer: refused, nothing freed
	  if (!detexGetErrorMessage() || !strstr(detexGetErrorMessage(), "was not returned")) { printf("detexhipFreePixelBuffer accepted a foreign pointer\n"); g_failures++; } }
	detexhipFreePixelBuffer(nullptr);
	REFUSED(detexhipCheckAbi(DETEXHIP_ABI_VERSION + 1) == 0);
	if (detexhipCheckAbi(DETEXHIP_ABI_VERSION) != 0) { printf("the header's own ABI version was refused\n"); g_failures++; }
	// mip levels
	detexhipLevel lv[17];
	for (auto &l : lv) l = detexhipLevel{ in, out, 16, 4, 4, 1, 1 };
	REFUSED(detexhipDecompressLevelsLinearDevice(BC1, lv, -1, RGBA8, nullptr, nullptr) == 0);
	REFUSED(detexhipDecompressLevelsLinearDevice(BC1, lv, 17, RGBA8, nullptr, nullptr) == 0);
	REFUSED(detexhipDecompressLevelsLinearDevice(BC1, nullptr, 2, RGBA8, nullptr, nullptr) == 0);
	REFUSED(detexhipDecompressLevelsLinearDevice(0x77000000u, lv, 1, RGBA8, nullptr, nullptr) == 0);
	lv[0].pitch_bytes = 8;
	REFUSED(detexhipDecompressLevelsLinearDevice(BC1, lv, 1, RGBA8, nullptr, nullptr) == 0);
	lv[0].pitch_bytes = 16; lv[0].width = -1;
	REFUSED(detexhipDecompressLevelsLinearDevice(BC1, lv, 1, RGBA8, nullptr, nullptr) == 0);
	// histogram
	uint32_t hist[16];
	REFUSED(detexhipModeHistogramDevice(0x55000000u, in, 1, hist, nullptr) == 0);
	REFUSED(detexhipModeHistogramDevice(BC1, in, 1, nullptr, nullptr) == 0);
	REFUSED(detexhipModeHistogramDevice(BC1, in + 3, 1, hist, nullptr) == 0);
	REFUSED(detexhipModeHistogramAccumulateDevice(BC1, in, (size_t)1 << 33, hist, nullptr) == 0);
	REFUSED(detexhipModeHistogram(0, in, 1, hist));
	// sharding and the multi-device entries
	int r0 = -7, r1 = -7;
	REFUSED(detexhipShardRows(8, 0, 0, &r0, &r1) == 0);
	REFUSED(detexhipShardRows(8, 4, 4, &r0, &r1) == 0);
	REFUSED(detexhipShardRows(-1, 4, 0, &r0, &r1) == 0);
	REFUSED(detexhipShardRows(8, 4, 0, nullptr, &r1) == 0);
	if (detexhipShardRows(0x7FFFFFFF, 64, 63, &r0, &r1) != 0 || r1 != 0x7FFFFFFF || r0 < 0 || r0 > r1) { printf("detexhipShardRows overflows\n"); g_failures++; }
	detexhipShard sh[2] = { { 0, nullptr, out, 0, 0, 0.f, 0, 0 }, { 0, nullptr, out, 0, 0, 0.f, 0, 0 } };
	float ms = 0;
	REFUSED(detexhipDecompressTextureLinearMultiDevice(BC1, in, 8, 8, 2, 2, 0, RGBA8, nullptr, 2, -1, nullptr, &ms, &ms) == 0);
	REFUSED(detexhipDecompressTextureLinearMultiDevice(BC1, in, 8, 8, 2, 2, 0, RGBA8, sh, 0, -1, nullptr, &ms, &ms) == 0);
	REFUSED(detexhipDecompressTextureLinearMultiDevice(BC1, in, 8, 8, 2, 2, 0, RGBA8, sh, 65, -1, nullptr, &ms, &ms) == 0);
	REFUSED(detexhipDecompressTextureLinearMultiDevice(BC1, in, 8, 8, 2, 2, 0, RGBA8, sh, 2, 0, nullptr, &ms, &ms) == 0);
	REFUSED(detexhipDecompressTextureLinearMultiDevice(BC1, in, 8, 8, 2, 2, 4, RGBA8, sh, 2, -1, nullptr, &ms, &ms) == 0);
	REFUSED(detexhipDecompressTextureLinearMultiDevice(0, in, 8, 8, 2, 2, 0, RGBA8, sh, 2, -1, nullptr, &ms, &ms) == 0);
	int devs[2] = { 0, 0 }, invalid = 0;
	REFUSED(detexhipDecompressTextureLinearMultiDeviceHost(BC1, in, 8, 8, 2, 2, out, 0, RGBA8, nullptr, 2, &invalid, &ms) == 0);
	REFUSED(detexhipDecompressTextureLinearMultiDeviceHost(BC1, nullptr, 8, 8, 2, 2, out, 0, RGBA8, devs, 2, &invalid, &ms) == 0);
	REFUSED(detexhipDecompressTextureLinearMultiDeviceHost(BC1, in, 8, 8, 2, 2, nullptr, 0, RGBA8, devs, 2, &invalid, &ms) == 0);
	REFUSED(detexhipDecompressTextureLinearMultiDeviceHost(BC1, in, 8, 8, 2, 2, out, 8, RGBA8, devs, 2, &invalid, &ms) == 0);
	REFUSED(detexhipDecompressTextureLinearMultiDeviceHost(BC1, in, 8, 8, 2, 2, out, 0, 0x7777u, devs, 2, &invalid, &ms) == 0);
	REFUSED(detexhipDecompressTextureLinearMultiDeviceHost(BC1, in, 8, -8, 2, 2, out, 0, RGBA8, devs, 2, &invalid, &ms) == 0);
	// device selection, knobs
	REFUSED(detexhipSetDevice(-1) == 0);
	REFUSED(detexhipSetDevice(1 << 20) == 0);
	detexhipSetKernelVariant(99); if (detexhipGetKernelVariant() != 0) { printf("variant 99 accepted\n"); g_failures++; }
	detexhipSetQuirks(0xFFFFFFFFu); if (detexhipGetQuirks() != DETEXHIP_QUIRKS_REFERENCE) { printf("quirk mask not clipped\n"); g_failures++; }
	detexhipSetQuirks(0); if (detexhipGetQuirks() != 0) { printf("quirk mask not stored\n"); g_failures++; }
	// the resident service's knob and counters (host_resident.cpp): clamped, previous value returned, NULL pointers accepted
	{
		const int before = detexhipSetResidentIdleMicroseconds(-5);
		if (detexhipSetResidentIdleMicroseconds(2000000000) != 0) { printf("negative idle time not clamped to 0\n"); g_failures++; }
		if (detexhipSetResidentIdleMicroseconds(before) != 1000000) { printf("idle time not clamped to one second\n"); g_failures++; }
		unsigned long long served = 1, started = 1;
		detexhipGetResidentStats(NULL, NULL);
		detexhipGetResidentStats(&served, &started);
		if (served != 0 || started != 0) { printf("resident counters of a thread that never decoded are not 0\n"); g_failures++; }
	}
	if (detexhipKernelName(0) != nullptr || detexhipKernelName(0xFF000000u) != nullptr || detexhipKernelName(BC1) == nullptr) { printf("detexhipKernelName\n"); g_failures++; }
	// the conversion-table builder over every half bit pattern (float -> int conversions under UBSan), monotone on [0, 1]
	unsigned prev = 0;
	for (uint32_t h = 0; h < 65536u; h++) {
		const unsigned v = detexhipHalfFloatToUNorm8((uint16_t)h);
		if (h <= 0x3C00u) { if (v < prev) { printf("half table not monotone at 0x%04X\n", h); g_failures++; break; } prev = v; }
		if (h > 0x8000u && h < 0xFC00u && v != 0) { printf("negative half 0x%04X -> %u\n", h, v); g_failures++; break; }
	}
	// the error convention with long and odd messages, and release without a context
	std::vector<char> big(100000, 'x'); big.back() = 0;
	detexSetErrorMessage("%s|%s", big.data(), big.data());
	if (!detexGetErrorMessage() || strlen(detexGetErrorMessage()) != 2 * (big.size() - 1) + 1) { printf("long error message truncated\n"); g_failures++; }
	detexSetErrorMessage("%%|%c|%5d|%-8s|", 'a', -3, "b");
	detexhipReleaseThreadResources();
	detexhipReleaseThreadResources();
	// WITH a device (the GPU box; tests/test_sanitized_host.py -m gpu): the host tier's real paths under the sanitizers -- the launch
	// per call, the resident service (requests, a format switch, an idle exit and restart, release while an instance lingers), a staged
	// texture -- each answer compared with the launch path's answer for the same input
	if (detexhipGetDeviceCount() > 0) {
		detexhipSetQuirks(DETEXHIP_QUIRKS_REFERENCE);
		std::vector<uint8_t> data(1 << 20), want(4 << 20), got(4 << 20);
		for (size_t k = 0; k < data.size(); k++) data[k] = (uint8_t)(k * 2654435761u >> 11);
		unsigned long long served0 = 0, served1 = 0;
		const uint32_t fmts[3] = { DETEX_TEXTURE_FORMAT_BC1, DETEX_TEXTURE_FORMAT_BPTC, DETEX_TEXTURE_FORMAT_BC3 };
		for (int side = 4; side <= 1024; side *= 4) {
			for (int f = 0; f < 3; f++) {
				detexTexture tx = { fmts[f], data.data(), side, side, side / 4, side / 4 };
				const size_t bytes = (size_t)side * side * 4;
				detexhipSetResidentIdleMicroseconds(0);			// a launch per call: the answer to compare with
				const bool ok0 = detexDecompressTextureLinear(&tx, want.data(), RGBA8);
				detexhipSetResidentIdleMicroseconds(400);
				for (int rep = 0; rep < 4; rep++) {
					memset(got.data(), 0xCD, bytes);
					const bool ok1 = detexDecompressTextureLinear(&tx, got.data(), RGBA8);
					if (ok0 != ok1 || memcmp(want.data(), got.data(), bytes) != 0) { printf("side %d format %d rep %d: answers differ\n", side, f, rep); g_failures++; }
					if (rep == 1) { struct timespec ts = { 0, 3000000 }; nanosleep(&ts, nullptr); }	// the instance leaves; the next request starts a new one
				}
				uint8_t px_a[64], px_b[64];
				if (f == 0) for (int rep = 0; rep < 4; rep++) {
					const bool oa = detexDecompressBlockBC1(data.data() + 8 * rep, DETEX_MODE_MASK_ALL, 0, rep == 0 ? px_a : px_b);
					if (!oa) { printf("one-block call failed: %s\n", detexGetErrorMessage()); g_failures++; }
				}
			}
		}
		detexhipGetResidentStats(&served1, nullptr);
		if (served1 - served0 < 20) { printf("the resident service answered only %llu requests\n", served1 - served0); g_failures++; }
		detexhipReleaseThreadResources();				// with an instance lingering
		printf("api_san: device part ran (%llu requests answered by resident kernels)\n", served1 - served0);
	}
	printf("api_san: %d problems, no sanitizer report\n", g_failures);
	return g_failures ? 2 : 0;
}
