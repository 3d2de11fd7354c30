cd /tmp && export TMPDIR=/tmp; export BIRDA_HIP_PRECISION=f16x3
for pref in "" "48,49,50,51,52,55,58,59,60"; do export BIRDA_HIP_MB_PREFER=$pref
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/cmp_y$pref -- python3 $GRAFT_REPO_ROOT/tools/gpu_quick_bench.py birdnet_v24 1000 1000 2>&1 | grep -E "iter 2|mbconv"; python3 $GRAFT_REPO_ROOT/tools/kstats.py "$GRAFT_REPO_ROOT/gpurun_out/cmp_y$pref" | grep -E "mbconv<3,1,(16|32),3|mbconv<5,1,(16|32),3|mbconv<5,1,(16|32),4|mbconv<5,2,(16|32),4"; done
