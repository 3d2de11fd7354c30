# parity + timing of the fused blocks after a change to the kernel skeleton (GPU box)
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "fused or full_size" 2>&1 | tail -3
export BIRDA_HIP_PRECISION=f16x3
cd /tmp && export TMPDIR=/tmp
BIRDA_HIP_MB_VERBOSE=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/persist -- python3 $GRAFT_REPO_ROOT/tools/gpu_quick_bench.py birdnet_v24 1000 1000 > $GRAFT_REPO_ROOT/gpurun_out/persist_bench.txt 2>&1
python3 $GRAFT_REPO_ROOT/tools/kstats.py $GRAFT_REPO_ROOT/gpurun_out/persist | grep mbconv | sort | awk '{printf "%s %s | ", $1, $5}'; echo
grep "iter 2" $GRAFT_REPO_ROOT/gpurun_out/persist_bench.txt
