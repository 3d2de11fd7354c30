cd /tmp && export TMPDIR=/tmp; export BIRDA_HIP_PRECISION=f16x3
for dbg in 0 1 2; do
  export BIRDA_HIP_MEL_DBG=$dbg
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/melabl_$dbg -- python3 $GRAFT_REPO_ROOT/tools/gpu_quick_bench.py birdnet_v24 1000 1000 > /dev/null 2>&1
  echo "== dbg $dbg: $(python3 $GRAFT_REPO_ROOT/tools/kstats.py $GRAFT_REPO_ROOT/gpurun_out/melabl_$dbg | grep mel_kernel | awk '{print $1,$2,$3,$6,$7}')"
done
