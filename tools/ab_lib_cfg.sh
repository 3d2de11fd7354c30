#!/bin/bash
# as ab_lib.sh, for one bench configuration and optional BIRDA_HIP_MB_PREFER lists for the new build:
#   bash tools/ab_lib_cfg.sh c4 "" "107,109"
cd ${GRAFT_REPO_ROOT:-.}
cfg=$1; shift
cp birda_amd/libbirda_hip.so /tmp/libbirda_hip_new.so
run() { python bench.py --config $cfg --no-cpu-baseline --no-extra-legs --steps ${STEPS:-8} --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); f=d['fused_block_us_per_1000_segments']
print('$1  %7.0f seg/s  mel %.3f  mbconv %.3f  %s' % (d['value'], d['stage_us_per_segment']['mel'], d['stage_us_per_segment']['mbconv'], ' '.join('%6.0f' % x for x in f.values())))"; }
for rep in 1 2; do
  cp tools/ab/libbirda_hip_old.so birda_amd/libbirda_hip.so; run old
  cp /tmp/libbirda_hip_new.so birda_amd/libbirda_hip.so
  for pref in "$@"; do BIRDA_HIP_MB_PREFER=$pref run "new[$pref]"; done
done
cp /tmp/libbirda_hip_new.so birda_amd/libbirda_hip.so
