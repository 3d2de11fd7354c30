#!/bin/bash
# host-fed legs of bench.py with the sub-slices of a slice in sequence (BIRDA_HIP_LANES=0) or on two streams, for several splits:
#   bash tools/ab_lanes.sh "0:0 1:0 1:4 1:6 1:8"      (LANES:SUBSLICES pairs; SUBSLICES 0 = the library's own split)
for pair in ${1:-0:0 1:0 1:4 1:6 1:8}; do
IFS=: read l n nl <<< "$pair"
BIRDA_HIP_LANES=$l BIRDA_HIP_SUBSLICES=$n BIRDA_HIP_NLANES=${nl:-2} python bench.py --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); h=d['h2d_inclusive']; e=d['end_to_end']
g=lambda x,k: round(x[k]['value']) if isinstance(x.get(k),dict) and 'value' in x[k] else None
print('LANES=$l SUB=$n NL=${nl:-2}', round(d['value']), {k:g(h,k) for k in h if g(h,k)}, {k:g(e,k) for k in e if g(e,k)})"
done
