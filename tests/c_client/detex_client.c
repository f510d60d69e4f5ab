/* tests/c_client/detex_client.c -- TEST-ONLY: a plain C program that uses libdetexhip the way the reference's own programs use
 * libdetex (validate.c:135,199-209: detexLoadKTXFile -> detexDecompressTextureLinear; detex.h:747-765, 826-836).  No Python, no
 * torch: compiled with gcc against a detex.h (this repository's, or the REFERENCE's own header where it is available at build
 * time -- the binary is the same client either way) and linked with -ldetexhip instead of -ldetex.  It prints one line per
 * input with the sha256 of the decoded pixels, which tests/test_c_client.py compares with the compiled reference's digests.
 *
 *   detex_client file.ktx ...            each file: load, decode into the format's native pixel format, digest
 *   detex_client --stream FORMAT BLOCKBYTES SEED W H     synthetic block stream U (splitmix64) of texture format word
 *                                                        FORMAT (detex.h:613-727), decoded and digested the same way
 *   detex_client --sha256-selftest       digest of "abc" and of 1,000,000 'a' (FIPS 180-4 vectors)
 *   detex_client --latency [owned]       microseconds per call (median of 2000) of detexDecompressBlockBC1 and of
 *                                        detexDecompressTextureLinear on 64x64 ... 4096x4096 BC1 and BC7 textures: what a C caller pays,
 *                                        without the ctypes overhead bench.py's host_tier_small carries; `owned`: the pixel buffers come from
 *                                        detexhipAllocPixelBuffer (pinned: the kernel writes straight into them)
 *   detex_client --oneshot[-breakdown] file.ktx ...   what a one-shot client pays (the reference's own callers decode their files once and
 *                                        exit: validate.c:188-223, detex-convert.c:310): milliseconds from main() to the first decoded
 *                                        texture and to the end of the whole file sequence, each file decoded ONCE into BGRA8 / BGRX8 like
 *                                        validate.c:199-209; -breakdown (libdetexhip builds) times the runtime's initialisation separately
 *                                        by asking for the device count first
 *   detex_client --blocks                n = 1, 1024, 1048576 independent blocks (BC1, BC7): the loop over the leaf function
 *                                        (detex.h:435-531) against ONE detexhipDecompressBlocks call (libdetexhip builds only:
 *                                        -DWITH_DETEXHIP), results compared block by block
 */
#define _POSIX_C_SOURCE 200809L
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "detex.h"
#ifdef WITH_DETEXHIP
#define DETEXHIP_COMPAT_DETEX_H		/* (a detex.h is already included: this repository's or the reference's) */
#include "detexhip.h"
#endif

/* ---- sha256 (FIPS 180-4), written for this test ------------------------------------------------------------------------------ */
static const uint32_t K256[64] = {
	0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3,
	0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
	0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
	0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
	0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
	0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2 };
typedef struct { uint32_t h[8]; uint64_t bytes; uint8_t buf[64]; size_t fill; } Sha256;
static uint32_t ror(uint32_t v, int s) { return (v >> s) | (v << (32 - s)); }
static void sha_block(Sha256 *c, const uint8_t *p) {
	uint32_t w[64], a[8];
	for (int i = 0; i < 16; i++) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
	for (int i = 16; i < 64; i++) {
		const uint32_t s0 = ror(w[i - 15], 7) ^ ror(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = ror(w[i - 2], 17) ^ ror(w[i - 2], 19) ^ (w[i - 2] >> 10);
		w[i] = w[i - 16] + s0 + w[i - 7] + s1;
	}
	memcpy(a, c->h, sizeof a);
	for (int i = 0; i < 64; i++) {
		const uint32_t t1 = a[7] + (ror(a[4], 6) ^ ror(a[4], 11) ^ ror(a[4], 25)) + ((a[4] & a[5]) ^ (~a[4] & a[6])) + K256[i] + w[i];
		const uint32_t t2 = (ror(a[0], 2) ^ ror(a[0], 13) ^ ror(a[0], 22)) + ((a[0] & a[1]) ^ (a[0] & a[2]) ^ (a[1] & a[2]));
		a[7] = a[6]; a[6] = a[5]; a[5] = a[4]; a[4] = a[3] + t1; a[3] = a[2]; a[2] = a[1]; a[1] = a[0]; a[0] = t1 + t2;
	}
	for (int i = 0; i < 8; i++) c->h[i] += a[i];
}
static void sha_init(Sha256 *c) {
	static const uint32_t h0[8] = { 0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19 };
	memcpy(c->h, h0, sizeof h0); c->bytes = 0; c->fill = 0;
}
static void sha_update(Sha256 *c, const uint8_t *p, size_t n) {
	c->bytes += n;
	if (c->fill) {
		const size_t take = 64 - c->fill < n ? 64 - c->fill : n;
		memcpy(c->buf + c->fill, p, take); c->fill += take; p += take; n -= take;
		if (c->fill == 64) { sha_block(c, c->buf); c->fill = 0; }
	}
	for (; n >= 64; p += 64, n -= 64) sha_block(c, p);
	if (n) { memcpy(c->buf, p, n); c->fill = n; }
}
static void sha_hex(Sha256 *c, char out[65]) {
	const uint64_t bits = c->bytes * 8;
	uint8_t pad[72] = { 0x80 };
	const size_t padlen = (c->fill < 56 ? 56 : 120) - c->fill;
	for (int i = 0; i < 8; i++) pad[padlen + i] = (uint8_t)(bits >> (56 - 8 * i));
	sha_update(c, pad, padlen + 8);
	for (int i = 0; i < 8; i++) sprintf(out + 8 * i, "%08x", c->h[i]);
}
static void sha256_of(const uint8_t *p, size_t n, char out[65]) { Sha256 c; sha_init(&c); sha_update(&c, p, n); sha_hex(&c, out); }

/* ---- the client --------------------------------------------------------------------------------------------------------------- */
static int decode_and_print(const char *label, const detexTexture *t) {
	const uint32_t pixel_format = detexGetPixelFormat(t->format);
	const size_t bytes = (size_t)t->width * (size_t)t->height * (size_t)detexGetPixelSize(pixel_format);
	uint8_t *pixels = (uint8_t *)malloc(bytes ? bytes : 1);
	if (!pixels) { printf("%s ERROR out of memory\n", label); return 1; }
	memset(pixels, 0xEE, bytes);
	const bool ok = detexDecompressTextureLinear(t, pixels, pixel_format);	/* the reference's call: texture.c:105, validate.c:208 */
	const char *message = detexGetErrorMessage();
	if (!ok && (!message || !strstr(message, "returned error"))) {		/* not "a block was invalid" but a failure of the library */
		printf("%s ERROR %s\n", label, message ? message : "(no message)");
		free(pixels);
		return 1;
	}
	char hex[65];
	sha256_of(pixels, bytes, hex);
	printf("%s format=0x%08X %dx%d ok=%d sha256=%s\n", label, t->format, t->width, t->height, ok ? 1 : 0, hex);
	free(pixels);
	return 0;
}

static double now_us(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e6 + t.tv_nsec * 1e-3; }
static int cmp_double(const void *a, const void *b) { const double x = *(const double *)a, y = *(const double *)b; return x < y ? -1 : x > y; }

/* owned != 0 (libdetexhip builds): the pixel buffers come from detexhipAllocPixelBuffer -- pinned, written by the kernel directly */
static int latency(int owned) {
	enum { N = 2000, WARM = 200 };
	static double t[N];
	uint8_t block[8] = { 0x12, 0x34, 0x56, 0x78, 0x9A, 0xBC, 0xDE, 0xF0 }, px[64], first[64];
	int wrong = 0;
	for (int i = -WARM; i < N; i++) {
		block[4] = (uint8_t)i;					/* another block every call: a stale answer would show */
		const double t0 = now_us();
		if (!detexDecompressBlockBC1(block, DETEX_MODE_MASK_ALL, 0, px)) { printf("latency ERROR %s\n", detexGetErrorMessage()); return 1; }
		if (i >= 0) t[i] = now_us() - t0;
		if (i == -WARM) memcpy(first, px, 64);
		else if (((i + WARM) & 255) == 0 && memcmp(first, px, 64) != 0) wrong++;	/* (block[4] is back at its first value every 256 calls) */
		else if (((i + WARM) & 255) == 1 && memcmp(first, px, 64) == 0) wrong++;
	}
	qsort(t, N, sizeof t[0], cmp_double);
	printf("latency one_block_us=%.2f p90=%.2f\n", t[N / 2], t[N * 9 / 10]);
	for (int f = 0; f < 2; f++) {
		const uint32_t format = f ? DETEX_TEXTURE_FORMAT_BPTC : DETEX_TEXTURE_FORMAT_BC1;
		const size_t bs = f ? 16 : 8;
		for (int side = 64; side <= 4096; side *= 2) {
			const int n = side <= 256 ? N : (side <= 1024 ? 300 : 40);	/* (the large ones take 0.1-25 ms per call on the CPU) */
			const int warm = side <= 1024 ? WARM : 8;
			const size_t nb = (size_t)(side / 4) * (side / 4), out_bytes = (size_t)side * side * 4;
			uint8_t *blocks = (uint8_t *)malloc(nb * bs), *expect = (uint8_t *)malloc(out_bytes), *pixels;
#ifdef WITH_DETEXHIP
			pixels = owned ? (uint8_t *)detexhipAllocPixelBuffer(out_bytes) : (uint8_t *)malloc(out_bytes);
#else
			pixels = (uint8_t *)malloc(out_bytes);
#endif
			if (!pixels) { printf("latency ERROR %s\n", detexGetErrorMessage()); return 1; }
			for (size_t k = 0; k < nb * bs; k++) blocks[k] = (uint8_t)(k * 2654435761u >> 13);
			if (f) for (size_t k = 0; k < nb; k++) blocks[k * bs] |= 1;		/* BC7: every block valid (mode 0), so the call returns true */
			detexTexture tex;
			tex.format = format; tex.data = blocks; tex.width = side; tex.height = side; tex.width_in_blocks = side / 4; tex.height_in_blocks = side / 4;
			for (int i = -warm; i < n; i++) {
				blocks[4] = (uint8_t)i;
				const double t0 = now_us();
				if (!detexDecompressTextureLinear(&tex, pixels, DETEX_PIXEL_FORMAT_RGBA8)) { printf("latency ERROR %s\n", detexGetErrorMessage()); return 1; }
				if (i >= 0) t[i] = now_us() - t0;
				if (i == -warm) memcpy(expect, pixels, out_bytes);
				else if (((i + warm) & 255) == 0 && memcmp(expect, pixels, out_bytes) != 0) wrong++;
				else if (((i + warm) & 255) == 1 && memcmp(expect, pixels, 64) == 0) wrong++;
			}
			qsort(t, n, sizeof t[0], cmp_double);
			printf("latency %s%s%dx%d_us=%.2f p90=%.2f\n", owned ? "owned_" : "", f ? "bc7_" : "", side, side, t[n / 2], t[n * 9 / 10]);
			free(blocks); free(expect);
#ifdef WITH_DETEXHIP
			if (owned) detexhipFreePixelBuffer(pixels); else free(pixels);
#else
			free(pixels);
#endif
		}
	}
	printf("latency wrong_results=%d\n", wrong);
	return wrong != 0;
}

/* validate.c:188-223 in a fresh process: every file loaded and decoded once (BGRA8 for formats with alpha, else BGRX8); the clock starts in main() */
static int oneshot(int argc, char **argv, int breakdown, double t_main) {
	double init_ms = -1.0, first_ms = -1.0, first_load_ms = -1.0;
	int decoded = 0, refused = 0;
	char hex[65] = "";
#ifdef WITH_DETEXHIP
	if (breakdown) { (void)detexhipGetDeviceCount(); init_ms = (now_us() - t_main) * 1e-3; }
#else
	(void)breakdown;
#endif
	for (int i = 0; i < argc; i++) {
		detexTexture *texture = NULL;
		if (!detexLoadKTXFile(argv[i], &texture)) { printf("oneshot ERROR %s: %s\n", argv[i], detexGetErrorMessage()); return 1; }
		if (i == 0) first_load_ms = (now_us() - t_main) * 1e-3;
		const uint32_t pixel_format = detexFormatHasAlpha(texture->format) ? DETEX_PIXEL_FORMAT_BGRA8 : DETEX_PIXEL_FORMAT_BGRX8;
		const size_t bytes = (size_t)texture->width * (size_t)texture->height * 4u;
		uint8_t *pixels = (uint8_t *)malloc(bytes);
		if (detexDecompressTextureLinear(texture, pixels, pixel_format)) decoded++; else refused++;	/* (validate.c:210-214 prints and goes on) */
		if (i == 0) { first_ms = (now_us() - t_main) * 1e-3; sha256_of(pixels, bytes, hex); }
		free(pixels); free(texture->data); free(texture);
	}
	const double all_ms = (now_us() - t_main) * 1e-3;
	printf("oneshot files=%d decoded=%d refused=%d init_ms=%.3f first_file_loaded_ms=%.3f first_call_ms=%.3f fixture_sequence_ms=%.3f first_sha256=%s\n", argc, decoded, refused,
		init_ms, first_load_ms, first_ms, all_ms, hex);
	return 0;
}

/* the migration of a per-block client: the loop over a leaf function against one batched call */
static int blocks_mode(void) {
	static const size_t counts[3] = { 1, 1024, 1048576 };
	int wrong = 0;
	for (int f = 0; f < 2; f++) {
		const uint32_t format = f ? DETEX_TEXTURE_FORMAT_BPTC : DETEX_TEXTURE_FORMAT_BC1;
		const size_t bs = f ? 16 : 8;
		(void)format;
		for (int c = 0; c < 3; c++) {
			const size_t n = counts[c];
			uint8_t *blocks = (uint8_t *)malloc(n * bs), *loop_px = (uint8_t *)malloc(n * 64), *loop_ok = (uint8_t *)malloc(n);
			uint64_t state = 0xB10C5 + 977 * c + f;
			for (size_t k = 0; k < n * bs; k += 8) {		/* splitmix64 */
				state += 0x9E3779B97F4A7C15ull;
				uint64_t z = state;
				z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
				memcpy(blocks + k, &z, 8);
			}
			double loop_us = -1.0, batched_us = -1.0;
#ifdef WITH_DETEXHIP
			const int do_loop = n <= 1024;			/* (a million trips to the GPU would take seconds) */
#else
			const int do_loop = 1;
#endif
			if (do_loop) {
				for (int rep = 0; rep < 3; rep++) {		/* best of three */
					const double t0 = now_us();
					for (size_t i = 0; i < n; i++) {
						loop_ok[i] = f ? detexDecompressBlockBPTC(blocks + i * bs, DETEX_MODE_MASK_ALL, 0, loop_px + i * 64)
							: detexDecompressBlockBC1(blocks + i * bs, DETEX_MODE_MASK_ALL, 0, loop_px + i * 64);
						if (!loop_ok[i]) memset(loop_px + i * 64, 0, 64);
					}
					const double dt = now_us() - t0;
					if (loop_us < 0 || dt < loop_us) loop_us = dt;
				}
			}
#ifdef WITH_DETEXHIP
			uint8_t *px = (uint8_t *)malloc(n * 64), *ok = (uint8_t *)malloc(n);
			for (int rep = 0; rep < 5; rep++) {
				memset(ok, 0xA5, n);
				const double t0 = now_us();
				const bool all = detexhipDecompressBlocks(format, blocks, n, DETEX_MODE_MASK_ALL, 0, px, ok);
				const double dt = now_us() - t0;
				if (rep && (batched_us < 0 || dt < batched_us)) batched_us = dt;		/* (the first call sizes the staging buffers) */
				int any_bad = 0;
				for (size_t i = 0; i < n; i++) any_bad |= ok[i] != 1;
				if (all == (any_bad != 0)) wrong++;
			}
			if (do_loop && (memcmp(px, loop_px, n * 64) != 0 || memcmp(ok, loop_ok, n) != 0)) wrong++;
			free(px); free(ok);
#endif
			printf("blocks format=%s n=%zu loop_us=%.2f batched_us=%.2f loop_ns_per_block=%.1f batched_ns_per_block=%.1f\n", f ? "BPTC" : "BC1", n, loop_us, batched_us,
				loop_us * 1e3 / (double)n, batched_us * 1e3 / (double)n);
			free(blocks); free(loop_px); free(loop_ok);
		}
	}
	printf("blocks wrong_results=%d\n", wrong);
	return wrong != 0;
}

int main(int argc, char **argv) {
	const double t_main = now_us();
	if (argc >= 3 && !strcmp(argv[1], "--oneshot")) return oneshot(argc - 2, argv + 2, 0, t_main);
	if (argc >= 3 && !strcmp(argv[1], "--oneshot-breakdown")) return oneshot(argc - 2, argv + 2, 1, t_main);
	if (argc >= 2 && !strcmp(argv[1], "--latency")) return latency(argc >= 3 && !strcmp(argv[2], "owned"));
	if (argc >= 2 && !strcmp(argv[1], "--blocks")) return blocks_mode();
	if (argc >= 2 && !strcmp(argv[1], "--sha256-selftest")) {
		char hex[65];
		sha256_of((const uint8_t *)"abc", 3, hex); printf("abc %s\n", hex);
		uint8_t *a = (uint8_t *)malloc(1000000); memset(a, 'a', 1000000);
		sha256_of(a, 1000000, hex); printf("million_a %s\n", hex);
		free(a);
		return 0;
	}
	if (argc == 7 && !strcmp(argv[1], "--stream")) {
		const uint32_t format = (uint32_t)strtoul(argv[2], NULL, 0), block_bytes = (uint32_t)strtoul(argv[3], NULL, 0);
		uint64_t state = strtoull(argv[4], NULL, 0);
		const int w = atoi(argv[5]), h = atoi(argv[6]);
		const size_t n_words = (size_t)(w / 4) * (size_t)(h / 4) * block_bytes / 8;
		uint64_t *words = (uint64_t *)malloc(n_words ? n_words * 8 : 8);
		if (!words) { printf("stream ERROR out of memory\n"); return 1; }
		for (size_t k = 0; k < n_words; k++) {			/* splitmix64 */
			state += 0x9E3779B97F4A7C15ull;
			uint64_t z = state;
			z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
			z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
			words[k] = z ^ (z >> 31);
		}
		detexTexture t;
		t.format = format; t.data = (uint8_t *)words; t.width = w; t.height = h; t.width_in_blocks = w / 4; t.height_in_blocks = h / 4;
		const int rc = decode_and_print("stream", &t);
		free(words);
		return rc;
	}
	int failures = 0;
	for (int i = 1; i < argc; i++) {
		detexTexture *texture = NULL;
		if (!detexLoadKTXFile(argv[i], &texture)) {		/* ktx.c:180, validate.c:135 */
			printf("%s ERROR %s\n", argv[i], detexGetErrorMessage() ? detexGetErrorMessage() : "(no message)");
			failures++;
			continue;
		}
		failures += decode_and_print(argv[i], texture);
		free(texture->data);					/* the caller owns both (ktx.c:150-176) */
		free(texture);
	}
	return failures ? 1 : 0;
}
