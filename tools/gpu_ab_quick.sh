#!/bin/bash
# quick same-run A/B against ab_libs/libdetexhip_prev.so:  bash tools/gpu_ab_quick.sh FORMATS STREAMS [linear|tiled|both]
FMTS=$1; STREAMS=$2; WHAT=${3:-both}
LIBS=${BASE_LIB:-ab_libs/libdetexhip_prev.so},detex_amd/lib/libdetexhip.so
for layout in linear tiled; do
  [ "$WHAT" = both ] || [ "$WHAT" = $layout ] || continue
  python tools/gpu_ab.py --libs $LIBS --formats $FMTS --streams $STREAMS --layout $layout --rounds 3 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$layout', d['lib'].split('/')[-1], d['format'], d['stream'], d['us'], d.get('rounds'))"
done
