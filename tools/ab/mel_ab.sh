cd $GRAFT_REPO_ROOT; cp birda_amd/libbirda_hip.so /tmp/keep.so
for rep in 1 2 3; do for l in old keep mel_1 mel_2 mel_6 mel_9; do
  if [ $l = keep ]; then cp /tmp/keep.so birda_amd/libbirda_hip.so; elif [ $l = old ]; then cp tools/ab/libbirda_hip_old.so birda_amd/libbirda_hip.so; else cp tools/ab/lib$l.so birda_amd/libbirda_hip.so; fi
  python bench.py --no-cpu-baseline --no-extra-legs --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('$l', round(d['value']), round(d['repeats']['median_of_5']), d['stage_us_per_segment']['mel'])"
done; done; cp /tmp/keep.so birda_amd/libbirda_hip.so
