#!/bin/bash
# Ablation sweep over ALL fused blocks (BIRDA_HIP_MB_DBG bits as in abl2.sh): us per 1000 segments of every block.
for dbg in ${ABL:-0 1 257 12 14 31 287 128 192 224}; do
  BIRDA_HIP_MB_DBG=$dbg python bench.py --no-cpu-baseline --no-extra-legs --steps 5 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); f=d['fused_block_us_per_1000_segments']; v=list(f.values())
print('dbg %3d  mbconv %.3f  %s' % ($dbg, d['stage_us_per_segment']['mbconv'], ' '.join('%6.0f' % x for x in v)))"
done
