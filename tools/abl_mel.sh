#!/bin/bash
# mel kernel ablations (BIRDA_HIP_MEL_DBG bits: 1 no main loop, 2 no power law): us per segment of the front-end stages
for dbg in ${ABL:-0 1 4 8 12 5 13 0}; do
  BIRDA_HIP_MEL_DBG=$dbg python bench.py --no-cpu-baseline --no-extra-legs --steps 8 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); s=d['stage_us_per_segment']
print('mel dbg %d  %7.0f seg/s  %s' % ($dbg, d['value'], ' '.join('%s %.3f' % kv for kv in s.items())))"
done
