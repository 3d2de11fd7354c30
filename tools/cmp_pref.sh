#!/bin/bash
# us per 1000 segments of every fused block under different BIRDA_HIP_MB_PREFER lists ("" = the planner's own choice), from
# bench.py's per-block HIP-event timing:  bash tools/cmp_pref.sh "" "85,87" ...   (A/B only within ONE gpurun call)
for pref in "$@"; do
  BIRDA_HIP_MB_PREFER=$pref python bench.py --no-cpu-baseline --no-extra-legs --steps ${STEPS:-8} --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); f=d['fused_block_us_per_1000_segments']
print('prefer %-12s %7.0f seg/s  mbconv %.3f' % ('\"$pref\"', d['value'], d['stage_us_per_segment']['mbconv']))
for k,v in f.items(): print('      %-48s %7.1f' % (k, v))"
done
