#!/bin/bash
# Ablation sweep of the fused blocks (BIRDA_HIP_MB_DBG bits: 1 no GELU in P1, 2 no P2, 4 no P3, 8 no P1 MFMA, 16 no weight DMA,
# 32 no output store, 64 no X loads, 128 no chunk loop).  Prints us per 1000 segments of the first five blocks.
for dbg in 0 128 192 224 64 16 32 2 4 8 1 15; do
  BIRDA_HIP_MB_DBG=$dbg python bench.py --no-cpu-baseline --no-extra-legs --steps 5 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); f=d['fused_block_us_per_1000_segments']; v=list(f.values())
print('dbg %3d  mbconv %.3f  first5 %s' % ($dbg, d['stage_us_per_segment']['mbconv'], ' '.join('%7.1f' % x for x in v[:5])))"
done
