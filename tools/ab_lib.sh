#!/bin/bash
# A/B of two builds of libbirda_hip.so within ONE gpurun call (boxes of the pool differ by +-4 %, and one box drifts by +-3 %
# between runs: take several alternations and read the medians):
#   build the other version, keep it as tools/ab/libbirda_hip_old.so (git-ignored, travels with gpurun), then on the box
#   REPS=5 STEPS=20 bash tools/ab_lib.sh [bench.py arguments]   -> per-block times old / new / old / new ..., then medians
cd ${GRAFT_REPO_ROOT:-.}
cp birda_amd/libbirda_hip.so /tmp/libbirda_hip_new.so
rm -f /tmp/ab_lib_rows.txt
for rep in $(seq 1 ${REPS:-2}); do for v in old new; do
  if [ $v = old ]; then cp tools/ab/libbirda_hip_old.so birda_amd/libbirda_hip.so; else cp /tmp/libbirda_hip_new.so birda_amd/libbirda_hip.so; fi
  python bench.py --no-cpu-baseline --no-extra-legs --steps ${STEPS:-8} --warmup 2 "$@" 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); f=d['fused_block_us_per_1000_segments']
print('$v  %7.0f seg/s  med5 %7.0f  mel %.3f  mbconv %.3f  %s' % (d['value'], d.get('repeats',{}).get('median_of_5',0), d['stage_us_per_segment']['mel'], d['stage_us_per_segment']['mbconv'], ' '.join('%6.0f' % x for x in f.values())))" | tee -a /tmp/ab_lib_rows.txt
done; done
cp /tmp/libbirda_hip_new.so birda_amd/libbirda_hip.so
python - <<'PY'
import statistics as st
rows={'old':[], 'new':[]}
for l in open('/tmp/ab_lib_rows.txt'):
    t=l.split(); rows[t[0]].append([float(t[1]), float(t[4]), float(t[6]), float(t[8])]+[float(x) for x in t[9:]])
for v in ('old','new'):
    if rows[v]:
        m=[st.median(c) for c in zip(*rows[v])]
        print('MEDIAN %s  %7.0f seg/s  med5 %7.0f  mel %.3f  mbconv %.3f  %s' % (v, m[0], m[1], m[2], m[3], ' '.join('%6.0f' % x for x in m[4:])))
PY
