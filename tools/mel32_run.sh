# A/B of the two front-end kernels under rocprofv3 (GPU box): bash tools/mel32_run.sh
export BIRDA_HIP_PRECISION=f16x3
cd /tmp && export TMPDIR=/tmp
for kind in birdnet_v24 perch_v2; do
  for m32 in 0 1; do
    export BIRDA_HIP_MEL32=$m32
    rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/mel32_${kind}_$m32 -- python3 $GRAFT_REPO_ROOT/tools/gpu_quick_bench.py $kind 600 600 > /dev/null 2>&1
    echo "== $kind BIRDA_HIP_MEL32=$m32: $(python3 $GRAFT_REPO_ROOT/tools/kstats.py $GRAFT_REPO_ROOT/gpurun_out/mel32_${kind}_$m32 | grep -E 'mel' | awk '{print $1,$2,$3,$4,$5,$6,$7}')"
  done
done
