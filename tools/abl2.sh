# skeleton ablations of the fused kernels: 63 = every phase off; +64 = no X loads; +128 = no chunk loop
cd /tmp && export TMPDIR=/tmp
for dbg in ${ABL_LIST:-63 127 191 255}; do
  export BIRDA_HIP_MB_DBG=$dbg
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/abl_$dbg -- python3 $GRAFT_REPO_ROOT/tools/gpu_quick_bench.py birdnet_v24 1000 1000 > /dev/null 2>&1
  echo "== dbg $dbg"; python3 $GRAFT_REPO_ROOT/tools/kstats.py $GRAFT_REPO_ROOT/gpurun_out/abl_$dbg | grep mbconv | sort | awk '{print $1, $5}' | tr '\n' ' '; echo
done
