#!/bin/bash
# small-launch twins of the two-segment tiles off / on (BIRDA_HIP_MB_TWIN): batch-256 / -512 rates and the host-fed legs of bench.py
for v in ${1:-0 1 0 1}; do
BIRDA_HIP_MB_TWIN=$v python bench.py --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); h=d['h2d_inclusive']; e=d['end_to_end']
print('TWIN=$v value', round(d['value']), 'b256', round(d['value_at_batch_256']), 'b512', round(d['value_at_batch_512']), 'pcm16_pinned', round(h['bh_predict_pcm16_pinned']['value']), 'pcm16', round(h['bh_predict_pcm16']['value']), 'contig_pinned', round(h['bh_predict_batch_contig_pinned']['value']), 'e2e', round(e['device']['value']), 'files', round(e['files_pipelined']['value']))"
done
