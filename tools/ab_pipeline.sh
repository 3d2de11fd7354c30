#!/bin/bash
# host-fed legs of bench.py (inside the bench process: the stream -> hardware-queue mapping depends on every stream the process has
# created) for several (lanes per context, packs in flight, sub-slices per pack):  bash tools/ab_pipeline.sh "3:3:2 2:3:2"
for cfg in ${1:-3:3:2 2:3:2 3:2:2}; do IFS=: read a b c <<< "$cfg"
BIRDA_HIP_NLANES=$a BIRDA_HOST_PIPELINE_DEPTH=$b BIRDA_HOST_PACK_SUBSLICES=$c python bench.py --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); h=d['h2d_inclusive']; e=d['end_to_end']
print('NLANES=$a DEPTH=$b PACKSUB=$c', round(h['bh_predict_pcm16_pinned']['value']), round(h['bh_predict_pcm16']['value']), round(h['bh_predict_batch_contig_pinned']['value']), round(e['device']['value']), e['files_pipelined']['runs'])"
done
