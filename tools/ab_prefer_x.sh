#!/bin/bash
# per-block times under tile-configuration preference lists in the EXPERIMENTS build (tools/ab/libbirda_hip_x.so, swapped in for the
# run; its kernels carry the phase clock, so compare within this script only):  bash tools/ab_prefer_x.sh "" "137" "137,139"
cd ${GRAFT_REPO_ROOT:-.}
cp birda_amd/libbirda_hip.so /tmp/libbirda_hip_new.so
cp tools/ab/libbirda_hip_x.so birda_amd/libbirda_hip.so
for rep in 1 2; do for pref in "$@"; do
  BIRDA_HIP_MB_PREFER=$pref python bench.py --no-cpu-baseline --no-extra-legs --steps ${STEPS:-8} --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); f=d['fused_block_us_per_1000_segments']
print('prefer %-10s %7.0f seg/s  mbconv %.3f  %s' % ('$pref', d['value'], d['stage_us_per_segment']['mbconv'], ' '.join('%6.0f' % x for x in list(f.values())[:6])))"
done; done
cp /tmp/libbirda_hip_new.so birda_amd/libbirda_hip.so
