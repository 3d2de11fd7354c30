#!/bin/bash
# A/B of two builds of libbirda_hip.so within ONE gpurun call (boxes of the pool differ by +-4 %):
#   build the other version, keep it as tools/ab/libbirda_hip_old.so (git-ignored, travels with gpurun), then on the box
#   bash tools/ab_lib.sh            -> per-block times old / new / old / new
cd ${GRAFT_REPO_ROOT:-.}
cp birda_amd/libbirda_hip.so /tmp/libbirda_hip_new.so
for rep in 1 2; do for v in old new; do
  if [ $v = old ]; then cp tools/ab/libbirda_hip_old.so birda_amd/libbirda_hip.so; else cp /tmp/libbirda_hip_new.so birda_amd/libbirda_hip.so; fi
  python bench.py --no-cpu-baseline --no-extra-legs --steps ${STEPS:-8} --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); f=d['fused_block_us_per_1000_segments']
print('$v  %7.0f seg/s  mel %.3f  mbconv %.3f  %s' % (d['value'], d['stage_us_per_segment']['mel'], d['stage_us_per_segment']['mbconv'], ' '.join('%6.0f' % x for x in f.values())))"
done; done
cp /tmp/libbirda_hip_new.so birda_amd/libbirda_hip.so
