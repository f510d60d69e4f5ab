// launchers.h -- host-side launch code of the decode kernels, instantiated per decoder by the formats_*.hip translation units
// (each includes its decoders' header, then this file, then defines its rows of the format table with FMT()).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstring>
#include <type_traits>

#include "host_internal.h"
#include "kernels.h"
#include "kernels_extra.h"
#include "kernels_resident.h"

namespace detexhip {

// Calls fn(std::integral_constant<int, EPI>) for the epilogue `epi` if the decoder's native pixel class can feed it
// (kernels.h: RGBA8-class natives take the R<->B swap and the RGB8 packing; 1/2-component natives and unsigned BC6H
// the three "to 8-bit RGB(X)" epilogues; BC6H also its own R<->B swap); hipErrorInvalidValue otherwise.
template <class Dec, class F> hipError_t with_epilogue(int epi, F &&fn) {
	constexpr int NC = NativeOf<Dec>::value;
	if (epi == kEpiNone) return fn(std::integral_constant<int, kEpiNone>{});
	if constexpr (NC == kNatRGBA8) {
		if (epi == kEpiSwapRB8) return fn(std::integral_constant<int, kEpiSwapRB8>{});
		if (epi == kEpiPackRGB8) return fn(std::integral_constant<int, kEpiPackRGB8>{});
	} else if constexpr (NC != kNatOther) {
		if constexpr (NC == kNatFloatRGBX16) {
			if (epi == kEpiSwapRB16) return fn(std::integral_constant<int, kEpiSwapRB16>{});
		}
		if (epi == kEpiToRGBX8) return fn(std::integral_constant<int, kEpiToRGBX8>{});
		if (epi == kEpiToBGRX8) return fn(std::integral_constant<int, kEpiToBGRX8>{});
		if (epi == kEpiToRGB8) return fn(std::integral_constant<int, kEpiToRGB8>{});
	}
	return hipErrorInvalidValue;
}

// decoders whose throughput kernels carry wave-uniform specialisations use the plain form in the kernels that
// are not on the throughput path (clipped geometry, mip levels), to bound code size (specialised in formats_bptc.hip)
template <class Dec> struct PlainDecoder { using type = Dec; };

// decode_linear's geometry: whole blocks, vector-aligned rows.  `sector_aligned`: every wave's 1 KiB store run also starts on
// a 64-byte boundary (row bytes, pitch and base multiples of 64) -- where it does not, the staged kernel is the faster one.
template <class Dec, int EPI> bool fast_geometry(const Geometry &g, bool *sector_aligned) {
	constexpr unsigned piece = 4u * EpilogueOf<Dec, EPI>::kRowDwords;
	constexpr unsigned align = piece % 16u == 0 ? 16u : (piece % 8u == 0 ? 8u : 4u);
	const uintptr_t place = reinterpret_cast<uintptr_t>(g.pixels) | (uintptr_t)g.pitch;
	*sector_aligned = ((place | ((uintptr_t)g.wb * piece)) & 63u) == 0;
	return (g.width & 3u) == 0 && (g.height & 3u) == 0 && g.wb * 4u == g.width && g.hb * 4u == g.height && (place % align) == 0;
}

// Dynamic LDS to request at launch so that at most `target` workgroups of `kernel` are resident per CU (0 = no cap).  The
// linear kernels are store-bound, and the write path runs better with FEWER concurrent store streams than the eight workgroups
// per CU the registers allow (BC1 8192^2 42.5 -> 41.5 us at three to five per CU, BC6H on coherent content 92.6 -> 84.7 at
// three); unused LDS is the one launch-time handle on residency.  The request is the CU's LDS (queried per device; 160 KiB on
// MI355X) divided by `target` and rounded down to 2 KiB, minus the kernel's static LDS -- and it is used only if the runtime's
// occupancy calculator then reports exactly `target` resident workgroups for it (checked once per kernel, device and target):
// on a device whose LDS size or allocation granule gives another answer the launch goes uncapped rather than wrongly capped.
inline size_t lds_bytes_per_cu(int device) {
	static std::atomic<size_t> cache[64];
	if (device < 0 || device >= 64) return 0;
	size_t v = cache[device].load(std::memory_order_relaxed);
	if (v == 0) {
		int bytes = 0;
		if (hipDeviceGetAttribute(&bytes, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, device) != hipSuccess || bytes <= 0) { (void)hipGetLastError(); bytes = 1; }
		v = (size_t)bytes;
		cache[device].store(v, std::memory_order_relaxed);
	}
	return v;
}
template <auto Kernel> unsigned occupancy_cap_lds(int target) {
	if (target < 3 || target > 7) return 0u;
	int device = 0;
	if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= 64) { (void)hipGetLastError(); return 0u; }
	// per device: (target + 1) << 32 | request -- a kernel is launched with ONE target in practice (its format's table entry), so one
	// slot per device suffices; another target simply recomputes
	static std::atomic<uint64_t> cache[64];
	const uint64_t known = cache[device].load(std::memory_order_relaxed);
	if ((known >> 32) == (uint64_t)target + 1u) return (unsigned)(known & 0xFFFFFFFFu);
	unsigned request = 0u;
	hipFuncAttributes attr{};
	if (hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(Kernel)) == hipSuccess) {
		// (the queried size first; 160 KiB, what gfx950 has, as the second candidate for runtimes that report the per-workgroup limit)
		const size_t candidates[2] = { lds_bytes_per_cu(device), (size_t)163840 };
		for (size_t per_cu : candidates) {
			const size_t per_workgroup = (per_cu / (size_t)target) & ~(size_t)2047;
			if (per_workgroup <= attr.sharedSizeBytes) continue;
			const unsigned dynamic = (unsigned)(per_workgroup - attr.sharedSizeBytes);
			int resident = 0;
			if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, Kernel, 256, dynamic) == hipSuccess && resident == target) { request = dynamic; break; }
		}
	}
	(void)hipGetLastError();
	cache[device].store(((uint64_t)target + 1u) << 32 | request, std::memory_order_relaxed);
	return request;
}
constexpr int workgroups_per_cu(int per_format) { return Tune::kWorkgroupsPerCu >= 0 ? Tune::kWorkgroupsPerCu : per_format; }

template <class Dec, int EPI> hipError_t launch_linear_epi(const Geometry &g) {
	const uint32_t n = g.wb * g.hb, tiles = (n + 255u) / 256u;
	uint8_t *px = static_cast<uint8_t *>(g.pixels);
	bool sector_aligned = false;
	// (formats with narrow rows keep the throughput kernels at any width: a 64-block run of R8 pixels is 256 bytes, and a
	// power-of-two width keeps those on sector boundaries anyway)
	constexpr bool kStagedWhenUnaligned = EpilogueOf<Dec, EPI>::kRowDwords >= 3;
	if (fast_geometry<Dec, EPI>(g, &sector_aligned) && (sector_aligned || !kStagedWhenUnaligned)) {
		// narrow pixels (RGTC1, SIGNED_RGTC1): several blocks per lane, so that a store instruction covers a longer run
		constexpr int kRow = EpilogueOf<Dec, EPI>::kRowDwords, kGroup = kRow * LaneBlocks<Dec>::value <= 4 ? LaneBlocks<Dec>::value : 1;
		if constexpr (kGroup > 1) {
			if (g.wb % kGroup == 0 && (reinterpret_cast<uintptr_t>(px) | g.pitch) % (4u * kRow * kGroup) == 0) {
				constexpr auto kernel = &decode_linear_grouped<Dec, EPI, true, kGroup>;
				hipLaunchKernelGGL(kernel, dim3((n / kGroup + 255u) / 256u), dim3(256), occupancy_cap_lds<kernel>(workgroups_per_cu(g.resident)), g.stream,
					g.blocks, px, g.wb, n, g.pitch, g.status, g.decode_flags);
				return hipGetLastError();
			}
		}
		// non-temporal row stores (43 vs 51 us with cached stores on BC1 8192^2)
		constexpr auto kernel = &decode_linear<Dec, EPI, true>;
		hipLaunchKernelGGL(kernel, dim3(tiles), dim3(256), occupancy_cap_lds<kernel>(workgroups_per_cu(g.resident)), g.stream, g.blocks, px, g.wb, n, g.pitch,
			g.status, g.decode_flags);
		return hipGetLastError();
	}
	// Everything else whose rows are dword-aligned and a whole number of dwords long -- clipped sizes (texture.c:116-120,
	// 132-136), rows that do not start on 64-byte boundaries -- goes through the staged kernel (kernels.h), one launch; rows
	// that are not even dword-aligned (R8 / RG8 / RGB8 images of odd width) pixel by pixel.
	using Plain = typename PlainDecoder<Dec>::type;
	const size_t row_bytes = (size_t)g.width * (EpilogueOf<Dec, EPI>::kRowDwords);		// width pixels * (4 * ROW / 4) bytes
	if (((reinterpret_cast<uintptr_t>(px) | (uintptr_t)g.pitch | row_bytes) & 3u) == 0 && row_bytes <= 0xFFFFFFFFull) {
		const uint32_t tiles_per_row = (g.wb + 255u) / 256u;
		constexpr auto kernel = &decode_linear_staged<Plain, EPI>;
		hipLaunchKernelGGL(kernel, dim3(tiles_per_row * g.hb), dim3(256), occupancy_cap_lds<kernel>(workgroups_per_cu(g.resident)), g.stream, g.blocks, px, g.wb,
			(uint32_t)row_bytes, g.height, g.pitch, g.status, tiles_per_row, g.decode_flags);
	} else {
		hipLaunchKernelGGL((decode_linear_clipped<Plain, EPI>), dim3(tiles), dim3(256), 0, g.stream, g.blocks, px, g.wb, n, g.width, g.height, g.pitch,
			g.status, g.decode_flags);
	}
	return hipGetLastError();
}

template <class Dec> hipError_t launch_linear(const Geometry &g) {
	if (g.wb * g.hb == 0) return hipSuccess;
	return with_epilogue<Dec>(g.epi, [&](auto epi) { return launch_linear_epi<Dec, decltype(epi)::value>(g); });
}

template <class Dec, int EPI> hipError_t launch_blocks_epi(const BatchArgs &a) {
	const uint32_t tiles = (uint32_t)((a.n + 255u) / 256u);
	uint8_t *px = static_cast<uint8_t *>(a.pixels);
	if (a.completion.done != nullptr) {		// host tier, small batch: everything in pinned host memory, completion word polled by the caller
		hipLaunchKernelGGL((decode_blocks_direct<typename PlainDecoder<Dec>::type, EPI>), dim3(tiles), dim3(256), 0, a.stream, a.blocks, px, (uint32_t)a.n,
			a.mode_mask, a.flags, a.ok, a.status, a.completion);
	} else if (a.checked) {
		hipLaunchKernelGGL((decode_blocks<typename PlainDecoder<Dec>::type, EPI, true>), dim3(tiles), dim3(256), 0, a.stream, a.blocks, px, (uint32_t)a.n,
			a.mode_mask, a.flags, a.ok, a.status);
	} else {
		constexpr auto kernel = &decode_blocks<Dec, EPI, false>;		// the block-major texture driver: store-bound like the linear kernel
		hipLaunchKernelGGL(kernel, dim3(tiles), dim3(256), occupancy_cap_lds<kernel>(workgroups_per_cu(a.resident)), a.stream, a.blocks, px, (uint32_t)a.n, a.mode_mask,
			a.flags, a.ok, a.status);
	}
	return hipGetLastError();
}

template <class Dec> hipError_t launch_blocks(const BatchArgs &a) {
	if (a.n == 0) return hipSuccess;
	return with_epilogue<Dec>(a.epi, [&](auto epi) { return launch_blocks_epi<Dec, decltype(epi)::value>(a); });
}

template <class Dec> hipError_t launch_single(const SingleArgs &a) {
	using Plain = typename PlainDecoder<Dec>::type;
	typename BlockWord<Dec::kBlockBytes>::type blk;
	memcpy(&blk, a.bitstring, sizeof blk);
	return with_epilogue<Dec>(a.epi, [&](auto epi) {
		hipLaunchKernelGGL((decode_single<Plain, decltype(epi)::value>), dim3(1), dim3(256), 0, a.stream, blk, a.mode_mask, a.flags, a.pixels, a.ok, a.done, a.ticket);
		return hipGetLastError();
	});
}

template <class Dec, int EPI> hipError_t launch_levels_epi(LevelsArgs &a) {
	constexpr unsigned row_bytes = 4u * EpilogueOf<Dec, EPI>::kRowDwords;
	constexpr unsigned align = row_bytes % 16u == 0 ? 16u : (row_bytes % 8u == 0 ? 8u : 4u);
	for (uint32_t l = 0; l < a.table.n_levels; l++) {
		LevelDesc &lv = a.table.level[l];
		const uint32_t hb = lv.width_in_blocks ? lv.n_blocks / lv.width_in_blocks : 0;
		lv.fast = (lv.width & 3u) == 0 && (lv.height & 3u) == 0 && lv.width_in_blocks * 4u == lv.width && hb * 4u == lv.height &&
			(reinterpret_cast<uintptr_t>(lv.pixels) % align) == 0 && (lv.pitch % align) == 0;
	}
	const uint32_t grid = a.table.wg_start[a.table.n_levels];
	if (grid == 0) return hipSuccess;
	hipLaunchKernelGGL((decode_levels<typename PlainDecoder<Dec>::type, EPI>), dim3(grid), dim3(256), 0, a.stream, a.table, a.status, a.decode_flags, a.completion);
	return hipGetLastError();
}
template <class Dec> hipError_t launch_levels(LevelsArgs &a) {
	return with_epilogue<Dec>(a.epi, [&](auto epi) { return launch_levels_epi<Dec, decltype(epi)::value>(a); });
}

template <class Dec> hipError_t launch_resident(const ResidentLaunch &r) {
	return with_epilogue<Dec>(r.epi, [&](auto epi) {
		hipLaunchKernelGGL((decode_resident<typename PlainDecoder<Dec>::type, decltype(epi)::value>), dim3(kResidentWorkgroups), dim3(256), 0, r.stream, r.args);
		return hipGetLastError();
	});
}

// one row of the format table.  RESIDENT / RESIDENT_BLOCKS: resident workgroups per CU of the linear kernels / of the block-major
// driver (occupancy_cap_lds), from the sweeps recorded in profiles/AB_RECORD.md (tools/gpu_wg_sweep.sh; 8192^2, streams U / C, caps
// 3..7 against none): the store-bound kernels with 32-bit or wider pixels and little VALU work gain 1-2.6 % at four or five per CU;
// BC6H gains 11 % on coherent content at five at the price of 2.6 % on uniform-random blocks; BC7, signed BC6H, ETC2_EAC (linear)
// and the narrow RGTC1 / EAC_R11 formats lose with any cap and keep what fits.
// RESIDENT_LARGE: the linear kernels' cap when blocks + pixels do not fit the 256 MiB Infinity Cache (-1: the same).  RGTC1 is the one
// format where the two regimes want different answers (same-run sweep, round 5: 16384^2 -- 384 MiB -- no cap 59.2 us, three to five
// per CU 55.3-55.5; 8192^2 -- 96 MiB, cache-resident -- no cap 12.4, capped 13.8-14.1): its stores are HBM-bound only when they reach HBM.
// READ_AHEAD: whether a texture of this format whose BLOCKS exceed the Infinity Cache is decoded in bands behind a read-only pass by default
// (detexhipSetReadAhead mode 1; device_tier.cpp).  Whole 32768^2 images, settled, one launch against the banded read-ahead
// (profiles/r06/footprint/sweep_32768_settled.jsonl): BC6H 1646 -> 1546 us (two other boxes 1680 -> 1490, 1706 -> 1505), BC1 782 -> 754
// (776 -> 749; 740 -> 752 on a third box) -- but BC3 783 -> 913, BC7 862 -> 948, ETC2_EAC 842 -> 901, signed BC6H 1595 -> 1618, ETC2 739 -> 744:
// a kernel whose mixed read + write stream already runs at the additive bound (writes at their rate + the blocks at the HBM peak) can only
// lose to a separate read pass at 5.7 TB/s.  So: BC6H and the BC1 pair, nothing else.
#define FMT_ROW(NAME, DEC, CLS, RESIDENT, RESIDENT_BLOCKS, RESIDENT_LARGE, READ_AHEAD) { #NAME, DETEX_TEXTURE_FORMAT_##NAME, &launch_linear<DEC>, &launch_blocks<DEC>, &launch_single<DEC>, \
	&launch_levels<DEC>, &launch_resident<DEC>, CLS, "decode_linear<detexhip::" #DEC, RESIDENT, RESIDENT_BLOCKS, RESIDENT_LARGE, READ_AHEAD }
#define FMT_L(NAME, DEC, CLS, RESIDENT, RESIDENT_BLOCKS, RESIDENT_LARGE) FMT_ROW(NAME, DEC, CLS, RESIDENT, RESIDENT_BLOCKS, RESIDENT_LARGE, false)
#define FMT(NAME, DEC, CLS, RESIDENT, RESIDENT_BLOCKS) FMT_L(NAME, DEC, CLS, RESIDENT, RESIDENT_BLOCKS, -1)
#define FMT_RA(NAME, DEC, CLS, RESIDENT, RESIDENT_BLOCKS) FMT_ROW(NAME, DEC, CLS, RESIDENT, RESIDENT_BLOCKS, -1, true)

}  // namespace detexhip
