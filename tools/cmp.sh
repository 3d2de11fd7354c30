cd /tmp && export TMPDIR=/tmp; export BIRDA_HIP_PRECISION=f16x3
for pref in "" "48,49,50,51,52,53"; do export BIRDA_HIP_MB_PREFER=$pref; export BIRDA_HIP_F16X3_ALL=$([ -n "$pref" ] && echo 1 || echo 0)
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/cmp_x$pref -- python3 $GRAFT_REPO_ROOT/tools/gpu_quick_bench.py birdnet_v24 1000 1000 2>&1 | grep -E "iter 2|mbconv"; python3 $GRAFT_REPO_ROOT/tools/kstats.py "$GRAFT_REPO_ROOT/gpurun_out/cmp_x$pref" | grep mbconv | head -8; done
