#!/bin/bash
# A/B of an environment switch within ONE box with per-fused-block times: bash tools/ab_blocks.sh VAR val1 val2 ...
# (each value twice, interleaved; us per 1 000 segments per block instantiation, from bench.py's HIP events)
var=$1; shift
for rep in 1 2; do for v in "$@"; do
  env $var=$v python bench.py --no-cpu-baseline --no-extra-legs --steps ${STEPS:-10} --warmup 3 ${BENCH_ARGS} 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); s=d['stage_us_per_segment']
print('$var=$v  %7.0f seg/s (median %7.0f)  mel %.3f  mbconv %.3f' % (d['value'], d['repeats']['median_of_5'], s['mel'], s['mbconv']))
print('    ' + ' '.join('%.0f' % v for v in d['fused_block_us_per_1000_segments'].values()))"
done; done
