cd /tmp && export TMPDIR=/tmp; export BIRDA_HIP_PRECISION=f16x3
for pref in "" "61,62,63"; do export BIRDA_HIP_MB_PREFER=$pref
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/cmp_z$pref -- python3 $GRAFT_REPO_ROOT/tools/gpu_quick_bench.py birdnet_v24 1000 1000 2>&1 | grep -E "iter 2|mbconv"; python3 $GRAFT_REPO_ROOT/tools/kstats.py "$GRAFT_REPO_ROOT/gpurun_out/cmp_z$pref" | grep -E "mbconv<3,2,16,1,|mbconv<3,1,16,1,|mbconv<3,1,16,2,"; done
