#!/bin/bash
# measurement builds of libdetexhip (never shipped): bash tools/build_exp_libs.sh nostore nocompute pad8192 sgprconst rgtc1g2 planarinlane bc7stage ...
set -e
cd "$(dirname "$0")/.."
for v in "$@"; do
  case $v in
    nostore) D=-DDETEXHIP_EXP_NOSTORE ;;
    nocompute) D=-DDETEXHIP_EXP_NOCOMPUTE ;;
    pad*) D=-DDETEXHIP_EXP_LDS_PAD=${v#pad} ;;
    waves*) D=-DDETEXHIP_EXP_BC7_WAVES=${v#waves} ;;
    persistent) D=-DDETEXHIP_EXP_BC7_PERSISTENT ;;             # BC7 (linear and block-major) on the persistent grid of round 2's first half
    plain) D=-DDETEXHIP_EXP_BC7_PLAIN ;;
    plain_persistent) D="-DDETEXHIP_EXP_BC7_PLAIN -DDETEXHIP_EXP_BC7_PERSISTENT" ;;
    sgprconst) D=-DDETEXHIP_EXP_SGPR_CONST ;;                 # v_bitop3 masks left in SGPRs (BC7 / BC6H)
    rgtc1g*) D=-DDETEXHIP_EXP_RGTC1_GROUP=${v#rgtc1g} ;;      # RGTC1 blocks per lane (1 = the one-block kernel)
    planar[0-9]*) D=-DDETEXHIP_EXP_PLANAR_SHARED=${v#planar} ;; # most planar blocks per wave decoded cooperatively (default 8)
    planarinlane) D=-DDETEXHIP_EXP_PLANAR_IN_LANE ;;          # ETC2 planar blocks always decoded in their own lanes
    bc7stage) D=-DDETEXHIP_EXP_BC7_SEPARATE_STAGE ;;          # block-major BC7 with the separate 17 KiB staging array
    *) D="$EXP_DEFS" ;;
  esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden $D -Wall -Wno-unused-function \
    -o detex_amd/lib/libdetexhip_exp_$v.so detex_amd/csrc/detexhip.hip detex_amd/csrc/ktx_loader.cpp &
done
wait
ls -la detex_amd/lib
