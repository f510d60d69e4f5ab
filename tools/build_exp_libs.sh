#!/bin/bash
# Measurement builds of libdetexhip into build/explib/ (never shipped with the product; the package directory holds the product library only):  bash tools/build_exp_libs.sh nostore nocompute prio1 ab_prio1 ...
# Each name maps to overrides of detex_amd/csrc/tune.h's ProductTune; a generated header carries them and is named in
# -DDETEXHIP_TUNE_HEADER (the one preprocessor switch of the measurement builds).  Names starting with ab_ also get
# the format-table translation units of tools/ab (the rejected A/B kernels, bench.py --variant N / DETEXHIP_VARIANT=N), so knobs and variants combine.
#   nostore / nocompute     decode without its stores / stores without the decode
#   wavesN                  BC7 register budget (waves per SIMD)
#   plain                   BC7 without the wave-uniform per-record copies
#   grpN                    BC7: subset rows requested together in the per-record copies
#   bc7stage                block-major BC7 with the separate 17 KiB staging array
#   bc6wavesN               BC6H register budget (waves per SIMD)
#   policyN / widepolicyN / blockspolicyN   cache policy of the linear kernels' row stores, pixels up to 32 bits / 64-bit pixels, and of the block-major kernels' stores: bit 0 sc0, bit 1 sc1,
#                           bit 2 nt (product: 6 = sc1 nt / 4 = nt)
#   rowwise                 BC6H linear kernel: texel rows exchanged and stored as the decoder completes them
#   staggerN                64-bit pixels, linear layout: wave w waits w * N * 64 cycles before its store burst (product: 4)
#   norowburst              64-bit pixels, linear layout: each texel row stored right after its exchange (round 3) instead of all eight stores in one burst
#   prioN / bc6prioN        s_setprio staging policy N of BC7 / BC6H (dev_common.h: stage_priority)
#   sgprconst               v_bitop3 masks left in SGPRs
#   rgtc1gN                 RGTC1 blocks per lane
#   hostduplexN             host tier: textures with at least N bytes of pixels upload and download at the same time (0 = never)
#   hostdirectN             host tier: byte threshold of the pinned-exchange path (0 = off)
#   hostpinnedinN           host tier: blocks of up to N bytes reach the staged path's kernel through the pinned buffer (0 = always uploaded)
#   rabandN                 read-ahead bands of at most N MiB of blocks (product: 128)
#   loadpolicyN             decode_linear: cache policy of the block load (bit 0 sc0, bit 1 sc1, bit 2 nt)
#   loadfirst               decode_linear: the block requested before the table copy's barrier
#   prefetchN               decode_linear: each wave also requests (and drops) the blocks N tiles further on
#   wgN                     N resident workgroups per CU for every linear kernel (0 = no cap; default: the per-format table)
#   sleepN                  s_sleep N between a wave's row stores (linear kernels)
#   planarN                 ETC2: most planar blocks per wave decoded cooperatively (0 = always in-lane)
#   several joined by '+':  nostore+prio1
set -e
cd "$(dirname "$0")/.."
mkdir -p build/tune build/explib
for v in "$@"; do
  body=""; srcs=""
  name=$v
  case $v in ab_*) srcs="formats_s3tc_rgtc_ab formats_etc_eac_ab formats_bptc_ab formats_bptc_float_ab histogram"; v=${v#ab_} ;; esac
  IFS='+' read -ra parts <<< "$v"
  for k in "${parts[@]}"; do
    case $k in
      base) ;;
      nostore) body+="static constexpr bool kNoStore = true; " ;;
      nocompute) body+="static constexpr bool kNoCompute = true; " ;;
      waves*) body+="static constexpr int kBc7WavesPerSimd = ${k#waves}; " ;;
      grp*) body+="static constexpr int kBc7UniformTexelGroup = ${k#grp}; " ;;
      plain) body+="static constexpr bool kBc7Uniform = false; " ;;
      bc7stage) body+="static constexpr bool kBc7OwnStage = false; " ;;
      prio*) body+="static constexpr int kBc7Prio = ${k#prio}; " ;;
      blockspolicy*) body+="static constexpr int kStorePolicyBlocks = ${k#blockspolicy}; " ;;
      widepolicy*) body+="static constexpr int kStorePolicyWide = ${k#widepolicy}; " ;;
      policy*) body+="static constexpr int kStorePolicy = ${k#policy}; " ;;
      rowwise) body+="static constexpr bool kRowWise = true; " ;;
      stagger*) body+="static constexpr int kWideStagger = ${k#stagger}; " ;;
      norowburst) body+="static constexpr bool kWideBurst = false; " ;;
      bc6waves*) body+="static constexpr int kBc6hWavesPerSimd = ${k#bc6waves}; " ;;
      bc6prio*) body+="static constexpr int kBc6hPrio = ${k#bc6prio}; " ;;
      sgprconst) body+="static constexpr bool kMasksInVgprs = false; " ;;
      rgtc1g*) body+="static constexpr int kRgtc1LaneBlocks = ${k#rgtc1g}; " ;;
      hostduplex*) body+="static constexpr unsigned long kHostDuplexBytes = ${k#hostduplex}; " ;;
      hostdirect*) body+="static constexpr unsigned long kHostDirectBytes = ${k#hostdirect}; " ;;
      hostpinnedin*) body+="static constexpr unsigned long kHostPinnedInputBytes = ${k#hostpinnedin}; " ;;
      raband*) body+="static constexpr unsigned long kReadAheadBandBytes = ${k#raband}ul << 20; " ;;
      loadpolicy*) body+="static constexpr int kLoadPolicy = ${k#loadpolicy}; " ;;
      loadfirst) body+="static constexpr bool kLoadBeforeTables = true; " ;;
      prefetch*) body+="static constexpr int kPrefetchTiles = ${k#prefetch}; " ;;
      wg*) body+="static constexpr int kWorkgroupsPerCu = ${k#wg}; " ;;
      sleep*) body+="static constexpr int kStoreSleep = ${k#sleep}; " ;;
      planar*) body+="static constexpr int kEtcPlanarShared = ${k#planar}; " ;;
      *) echo "unknown knob $k" >&2; exit 2 ;;
    esac
  done
  hdr=$PWD/build/tune/tune_$name.h
  echo "struct Tune : ProductTune { $body};" > "$hdr"
  # (one sub-make per build: its own object directory, the library's translation units compiled in parallel)
  make -s -j4 lib LIB=build/explib/libdetexhip_exp_$name.so OBJDIR=build/obj_exp/$name ${srcs:+SRCS_HIP="$srcs"} EXTRA_HIPFLAGS="-DDETEXHIP_TUNE_HEADER='\"$hdr\"'" &
done
wait
ls -la build/explib
