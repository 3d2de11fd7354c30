for f in 0 64 32 96 0 64; do
BIRDA_HIP_FIRST_SUBSLICE=$f python bench.py --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); h=d['h2d_inclusive']; e=d['end_to_end']
print('FIRST=$f', round(h['bh_predict_pcm16_pinned']['value']), round(h['bh_predict_pcm16']['value']), round(h['bh_predict_batch_contig_pinned']['value']), round(e['device']['value']), round(e['files_pipelined']['value']))"
done
