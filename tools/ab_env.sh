#!/bin/bash
# A/B of an environment switch within ONE box: bash tools/ab_env.sh VAR val1 val2 ...  (each value twice, interleaved)
var=$1; shift
for rep in 1 2; do for v in "$@"; do
  env $var=$v python bench.py --no-cpu-baseline --no-extra-legs --steps ${STEPS:-8} --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); s=d['stage_us_per_segment']
print('$var=$v  %7.0f seg/s  %s' % (d['value'], ' '.join('%s %.3f' % kv for kv in s.items() if kv[1] > 0)))"
done; done
