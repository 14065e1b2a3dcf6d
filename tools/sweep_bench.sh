#!/bin/bash
# development: bench.py over a list of crt_tuning strings (one per line on stdin); prints value / ms / phase times per line
while IFS= read -r T; do
  [ "$T" = "#" ] && continue
  python bench.py --scene ${SCENE:-hw14} --steps 20 --warmup 5 --no-cpu-baseline --no-alone --in-flight 0 --tuning "$T" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('%-60s %7.1f Mpx/s %6.3f ms  levels %.3f shadow0 %.3f tail %.3f  ok=%s fb=%d' % (sys.argv[1], d['value'], d['ms_per_step'], d['kernel_ms']['recursion_levels'], d['kernel_ms']['shadow_pass0_overlapped'], d['kernel_ms']['shadow_pass1_heavy_resolve'], d['frame_matches_counting_build'], d['fallback_frames']))
" "$T"
done
