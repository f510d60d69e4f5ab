// tune.h -- the handful of constants the measurement builds of profiles/AB_RECORD.md vary.
//
// The product library is compiled with ProductTune, and this file holds the ONLY preprocessor switch of the
// measurement builds: tools/build_exp_libs.sh writes a small header that derives `Tune` from ProductTune with one or two
// members overridden and names it in DETEXHIP_TUNE_HEADER.  Kernels and decoders read `Tune::k...` in `if constexpr`
// / template arguments, so the product translation unit carries no measurement `#if`.
#pragma once

namespace detexhip {

struct ProductTune {
	// decode without its stores / stores without the decode (profiles/*/compute_vs_memory.jsonl)
	static constexpr bool kNoStore = false, kNoCompute = false;
	// BC7: register budget in waves per SIMD (80 VGPRs, no spill), the wave-uniform per-record copies of the decoder, the
	// block-major exchange staged in the decoder's own (dead) lane rows, and the s_setprio staging of the decode (0 = off,
	// 1 = priority rises with the wave's progress)
	static constexpr int kBc7WavesPerSimd = 6;
	static constexpr bool kBc7Uniform = true;
	static constexpr int kBc7UniformTexelGroup = 4;	// subset rows requested together in the per-record copies (1 in the mixed-mode path)
	static constexpr bool kBc7OwnStage = true;
	static constexpr int kBc7Prio = 0;
	// BC6H: the same priority staging; register budget in waves per SIMD (0 = the compiler's choice)
	static constexpr int kBc6hPrio = 0;
	static constexpr int kBc6hWavesPerSimd = 0;
	// BC6H linear kernel: texel rows exchanged and stored as the decoder completes them (false: after the whole block -- the
	// faster way round: profiles/AB_RECORD.md)
	static constexpr bool kRowWise = false;
	// 64-bit pixels, linear layout: a wave's four texel rows all exchanged through LDS before its eight stores are issued (one burst)
	static constexpr bool kWideBurst = true;
	static constexpr int kWideStagger = 4;		// ... wave w of the workgroup waiting w * this * 64 cycles before its burst (0 = none)
	// resident workgroups per CU of the linear kernels: -1 = the per-format choice of the formats_*.hip tables, 0 = no cap, 3..7 = this
	// many for every format (sweeps; the cap is dynamic LDS requested at launch: profiles/AB_RECORD.md)
	static constexpr int kWorkgroupsPerCu = -1;
	// memory-side cache of the device (MI355X: 256 MiB Infinity Cache): a texture whose blocks + pixels fit is not HBM-bound
	static constexpr unsigned long kInfinityCacheBytes = 256ul << 20;
	// Textures whose compressed blocks alone exceed that cache (32768^2 BC1: 512 MiB) are decoded in bands of block rows whose blocks fit
	// this many bytes, each band's blocks first read into the cache by a read-only pass (histogram.hip: read_ahead) and then decoded:
	// HBM sees a read phase and a write phase instead of reads scattered through the write stream (device_tier.cpp: linear_device_with;
	// DESIGN.md section 4, "the large-footprint cliff")
	static constexpr unsigned long kReadAheadBandBytes = 128ul << 20;
	// decode_linear: every wave also requests the blocks of the tile this many tiles further on (a multiple of eight: the same XCD's L2) and
	// never uses them -- a read-ahead inside the launch for blocks that come out of HBM (0 = none; measurement: profiles/AB_RECORD.md round 6)
	static constexpr int kPrefetchTiles = 0;
	// decode_linear: the block requested BEFORE the format tables are copied into LDS (and waited for behind that copy's barrier) instead of
	// after it (measurement builds; the product loads after the barrier: profiles/AB_RECORD.md rounds 2 and 6)
	static constexpr bool kLoadBeforeTables = false;
	// decode_linear: cache policy of the block load (bits as for the stores; 0 = the compiler's plain load)
	static constexpr int kLoadPolicy = 0;
	// s_sleep argument between a wave's row stores (0 = none): does a smoother store issue raise the write rate? (profiles/AB_RECORD.md)
	static constexpr int kStoreSleep = 0;
	// cache policy of the row stores of the linear kernels (bit 0 sc0, bit 1 sc1, bit 2 nt; 4 = what __builtin_nontemporal_store
	// emits): pixels of up to 32 bits / the 64-bit pixels of BC6H (kernels.h: store_with_policy)
	static constexpr int kStorePolicy = 6;
	static constexpr int kStorePolicyWide = 4;
	static constexpr int kStorePolicyBlocks = 6;	// the block-major kernels' 16-byte stores
	// v_bitop3_b32 masks pinned into VGPRs (an SGPR source halves the issue rate of a full-rate VALU op)
	static constexpr bool kMasksInVgprs = true;
	// RGTC1: blocks per lane in the linear kernel
	static constexpr int kRgtc1LaneBlocks = 4;
	// host-pointer tier: textures whose blocks + pixels fit in this many bytes are exchanged through pinned host memory the
	// kernel reads and writes directly (one launch + one synchronisation; host_tier.cpp: direct_exchange)
	static constexpr unsigned long kHostDirectBytes = 1280u << 10;	// up to 512 x 512 RGBA8 (measured: 64^2 51 -> 18 us, 256^2 83 -> 31 us per call)
	// ... and larger textures whose BLOCKS fit in this many bytes hand them over through the same pinned buffer (read by the kernel across
	// the link) instead of an upload out of pageable memory; 0 = always upload
	static constexpr unsigned long kHostPinnedInputBytes = 1024u << 10;
	// ... and a pixel buffer the library handed out (detexhipAllocPixelBuffer) is written by the kernel directly up to this many bytes of pixels
	static constexpr unsigned long kOwnedDirectBytes = 8ul << 20;
	// ... and textures with at least this many bytes of blocks AND of pixels go up and come down AT THE SAME TIME: bands of block rows, band k + 1 uploaded by a
	// helper thread while band k is decoded and downloaded (the link is full duplex; copies between pageable memory and the device block their
	// calling thread, hence the second thread: host_tier.cpp: via_staging_duplex); 0 = never
	static constexpr unsigned long kHostDuplexBytes = 32ul << 20;
	static constexpr int kHostDuplexBands = 8;
	// ETC2: most planar blocks per wave that are decoded cooperatively (0 = always in their own lanes)
	static constexpr int kEtcPlanarShared = 8;
};

#if defined(DETEXHIP_TUNE_HEADER)
#include DETEXHIP_TUNE_HEADER	// measurement build: `struct Tune : ProductTune { ...overrides... };`
#else
using Tune = ProductTune;
#endif

}  // namespace detexhip
