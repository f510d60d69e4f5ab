#!/bin/bash
# Round 3, GPU call B: parity of the new clipped path / BC7 changes, fuzz slice, BC7 A/B, clipped timings, fill references.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r03b
rm -rf $OUT; mkdir -p $OUT
E=build/explib
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider --durations=8 2>&1 | tail -16 | tee $OUT/pytest_gpu.log
echo "== fuzz 40 s"; timeout 300 python tools/gpu_fuzz.py 40 5000 2>&1 | tail -3 | tee $OUT/fuzz.log
echo "== BC7 A/B"
timeout 900 python tools/gpu_ab.py --libs $PWD/detex_amd/lib/libdetexhip.so,$E/libdetexhip_g1.so,$E/libdetexhip_r02.so --formats BPTC --streams U,M,C --rounds 3 --out $OUT/bc7_ab.jsonl 2>>$OUT/ab.err | cut -c1-200
timeout 900 python tools/gpu_ab.py --libs $PWD/detex_amd/lib/libdetexhip.so,$E/libdetexhip_g1.so,$E/libdetexhip_r02.so --formats BPTC --streams U,C --layout tiled --rounds 2 --out $OUT/bc7_tiled.jsonl 2>>$OUT/ab.err | cut -c1-200
echo "== other formats unchanged? (product vs round-2 library)"
timeout 900 python tools/gpu_ab.py --libs $PWD/detex_amd/lib/libdetexhip.so,$E/libdetexhip_r02.so --formats BC1,BC3,ETC2_EAC,RGTC1,BPTC_FLOAT,BPTC_SIGNED_FLOAT --streams U --rounds 2 --out $OUT/others_ab.jsonl 2>>$OUT/ab.err | cut -c1-200
echo "== clipped geometry"; timeout 300 python tools/gpu_clipped_timing.py 2>>$OUT/ab.err | tee $OUT/clipped.txt
echo "== fill references"; timeout 600 python tools/gpu_hbm_ref.py $OUT/hbm_reference.jsonl 2>$OUT/hbm.err | grep -E "fill_image|dword_aligned|torch" | cut -c1-170
tail -3 $OUT/ab.err
echo "== done"
