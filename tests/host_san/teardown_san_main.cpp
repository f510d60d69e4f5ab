// tests/host_san/teardown_san_main.cpp -- TEST-ONLY: how the host tier's per-thread state dies.  Every host-pointer thread owns a
// thread_local context (a stream, staging buffers, a resident service kernel that may still be running) whose destructor calls into the
// HIP runtime at thread exit and at process exit (host_tier.cpp: ThreadContext, multi_device.cpp: ShardSlots).  This program, built with
// AddressSanitizer + UndefinedBehaviorSanitizer on the host code (make teardown-san), lets threads decode and EXIT WITHOUT
// detexhipReleaseThreadResources() while their resident kernels are alive, then ends the process in one of three ways:
//   teardown_san threads            linked against the library's objects; the main thread decodes too and returns from main() with its
//                                   resident kernel still lingering (idle time 200 ms)
//   teardown_san exit               the same, but leaves through exit() from inside a worker thread while the others have finished
//   teardown_san dlclose LIB.so     the instrumented library dlopen()ed, used by four threads that exit, dlclose()d, opened and used again
// Needs a HIP device (tests/test_sanitized_host.py -m gpu); prints "teardown_san: ok" and returns 0, or a sanitizer report / crash.
#include <dlfcn.h>
#include <pthread.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/detex.h"
#include "../../include/detexhip.h"

struct Api {
	bool (*linear)(const detexTexture *, uint8_t *, uint32_t);
	bool (*block_bc1)(const uint8_t *, uint32_t, uint32_t, uint8_t *);
	int (*set_idle)(int);
	void (*stats)(unsigned long long *, unsigned long long *);
	const char *(*error)(void);
	int (*device_count)(void);
};
static Api g_api;
static int g_failures = 0;

struct Work { int id, calls; uint8_t first_pixels[64 * 64 * 4]; bool ok; };

static void *worker(void *arg) {
	Work *w = static_cast<Work *>(arg);
	std::vector<uint8_t> blocks(16 * 16 * 8), pixels(64 * 64 * 4);
	for (size_t k = 0; k < blocks.size(); k++) blocks[k] = (uint8_t)((k + 7u * (unsigned)w->id) * 2654435761u >> 9);
	detexTexture t = { DETEX_TEXTURE_FORMAT_BC1, blocks.data(), 64, 64, 16, 16 };
	w->ok = true;
	for (int i = 0; i < w->calls; i++) {
		memset(pixels.data(), 0xCD, pixels.size());
		if (!g_api.linear(&t, pixels.data(), DETEX_PIXEL_FORMAT_RGBA8)) { printf("thread %d call %d: %s\n", w->id, i, g_api.error()); w->ok = false; break; }
		if (i == 0) memcpy(w->first_pixels, pixels.data(), pixels.size());
		else if (memcmp(w->first_pixels, pixels.data(), pixels.size()) != 0) { printf("thread %d call %d: answers differ\n", w->id, i); w->ok = false; break; }
		uint8_t px[64];
		if (!g_api.block_bc1(blocks.data() + 8 * (i % 200), DETEX_MODE_MASK_ALL, 0, px)) { printf("thread %d: one-block call failed\n", w->id); w->ok = false; break; }
	}
	unsigned long long served = 0, started = 0;
	g_api.stats(&served, &started);
	if (w->ok && served < (unsigned long long)w->calls / 2) { printf("thread %d: only %llu requests went to a resident kernel\n", w->id, served); w->ok = false; }
	return nullptr;		// NO detexhipReleaseThreadResources(): the thread_local destructors do it, with the resident kernel still running
}

static int run_threads(int n, int calls) {
	std::vector<pthread_t> th(n);
	std::vector<Work> work(n);
	for (int k = 0; k < n; k++) { work[k].id = k; work[k].calls = calls; pthread_create(&th[k], nullptr, worker, &work[k]); }
	int bad = 0;
	for (int k = 0; k < n; k++) { pthread_join(th[k], nullptr); bad += work[k].ok ? 0 : 1; }
	return bad;
}

static void *exiting_worker(void *arg) {
	worker(arg);
	printf("teardown_san: ok (exit() from a worker thread)\n");
	fflush(stdout);
	exit(g_failures ? 2 : 0);		// the main thread is blocked in pthread_join; its context and this thread's are both alive
}

template <class F> static void sym(void *h, const char *name, F &fn) {
	fn = reinterpret_cast<F>(dlsym(h, name));
	if (!fn) { printf("dlsym(%s): %s\n", name, dlerror()); exit(3); }
}
static void *open_library(const char *path) {
	void *h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
	if (!h) { printf("dlopen(%s): %s\n", path, dlerror()); exit(3); }
	sym(h, "detexDecompressTextureLinear", g_api.linear); sym(h, "detexDecompressBlockBC1", g_api.block_bc1);
	sym(h, "detexhipSetResidentIdleMicroseconds", g_api.set_idle); sym(h, "detexhipGetResidentStats", g_api.stats);
	sym(h, "detexGetErrorMessage", g_api.error); sym(h, "detexhipGetDeviceCount", g_api.device_count);
	return h;
}

int main(int argc, char **argv) {
	const char *mode = argc > 1 ? argv[1] : "threads";
	if (!strcmp(mode, "dlclose")) {
		if (argc < 3) { printf("usage: teardown_san dlclose LIB.so\n"); return 3; }
		for (int round = 0; round < 2; round++) {
			void *h = open_library(argv[2]);
			if (g_api.device_count() <= 0) { printf("teardown_san: no HIP device\n"); return 4; }
			g_api.set_idle(200000);
			g_failures += run_threads(4, 60);
			Work mine; mine.id = 9; mine.calls = 30;
			worker(&mine);					// the main thread's own context stays alive across the dlclose
			g_failures += mine.ok ? 0 : 1;
			if (dlclose(h) != 0) { printf("dlclose: %s\n", dlerror()); g_failures++; }
		}
		printf("teardown_san: %s (dlclose)\n", g_failures ? "FAILED" : "ok");
		return g_failures ? 2 : 0;
	}
#ifdef TEARDOWN_SAN_LINKED
	g_api = Api{ detexDecompressTextureLinear, detexDecompressBlockBC1, detexhipSetResidentIdleMicroseconds, detexhipGetResidentStats, detexGetErrorMessage, detexhipGetDeviceCount };
	if (g_api.device_count() <= 0) { printf("teardown_san: no HIP device\n"); return 4; }
	g_api.set_idle(200000);				// 200 ms: every context dies with its resident kernel running
	g_failures += run_threads(4, 80);
	Work mine; mine.id = 8; mine.calls = 40;
	worker(&mine);
	g_failures += mine.ok ? 0 : 1;
	if (!strcmp(mode, "exit")) {
		pthread_t t; Work w; w.id = 5; w.calls = 40;
		pthread_create(&t, nullptr, exiting_worker, &w);
		pthread_join(t, nullptr);
		return 5;				// not reached
	}
	printf("teardown_san: %s (return from main with a resident kernel lingering)\n", g_failures ? "FAILED" : "ok");
	return g_failures ? 2 : 0;
#else
	printf("teardown_san: built without the library linked in; use the dlclose mode\n");
	return 3;
#endif
}
