cd /tmp && export TMPDIR=/tmp
export BIRDA_HIP_MB_PREFER=11,12,13,3,4,16,6,7,19
for dbg in 0 1 2 4 8 16 32 63; do
  export BIRDA_HIP_MB_DBG=$dbg
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/abl_$dbg -- python3 $GRAFT_REPO_ROOT/tools/gpu_quick_bench.py birdnet_v24 1024 1024 > /dev/null 2>&1
  echo "== dbg $dbg"; python3 $GRAFT_REPO_ROOT/tools/kstats.py $GRAFT_REPO_ROOT/gpurun_out/abl_$dbg | grep mbconv | sort | awk '{print $1, $5}' | tr '\n' ' '; echo
done
