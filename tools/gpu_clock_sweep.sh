set -u
run() { python tools/gpu_ab.py --libs detex_amd/lib/libdetexhip.so --formats BPTC_FLOAT,BC1,BPTC --streams Z,C,U --rounds 1 --clocks 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); c=d['clocks']; print('   ', d['format'], d['stream'], d['us'], 'us', c['sclk_MHz_mean'], 'MHz', c['W_mean'], 'W')"; }
echo "== default"; run
echo "== setsrange 500 1800"; rocm-smi --setsrange 500 1800 --autorespond y 2>&1 | grep -v "^=\|^$" | head -4; run
echo "== setsrange 500 1500"; rocm-smi --setsrange 500 1500 --autorespond y 2>&1 | grep -v "^=\|^$" | head -4; run
echo "== setsrange 500 2100"; rocm-smi --setsrange 500 2100 --autorespond y 2>&1 | grep -v "^=\|^$" | head -4; run
rocm-smi --setsrange 500 2400 --autorespond y 2>&1 | grep -v "^=\|^$" | head -2
echo "== power cap 1000 W"; rocm-smi --setpoweroverdrive 1000 --autorespond y 2>&1 | grep -v "^=\|^$" | head -4; run
rocm-smi --resetpoweroverdrive --autorespond y 2>&1 | grep -v "^=\|^$" | head -2
