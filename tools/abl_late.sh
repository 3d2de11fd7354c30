#!/bin/bash
# ablations of the late fused blocks (BIRDA_HIP_MB_DBG bits: 2 no depthwise phase, 4 no project MFMAs, 8 no expand MFMAs, 16 no weight
# DMA, 32 no stores, 64 no X loads, 128 no chunk loop), us per 1 000 segments per block:  bash tools/abl_late.sh [prefer list]
for dbg in 0 32 16 2 4 8 64 128 0; do
  BIRDA_HIP_MB_PREFER=$1 BIRDA_HIP_MB_DBG=$dbg python bench.py --no-cpu-baseline --no-extra-legs --steps ${STEPS:-6} --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); f=d['fused_block_us_per_1000_segments']
print('dbg %3d  ' % $dbg + ' '.join('%6.0f' % v for k, v in f.items() if ',32,' in k[:16]))"
done
