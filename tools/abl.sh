# Ablations of the fused blocks (BIRDA_HIP_MB_DBG bits: 2 no depthwise phase, 4 no project GEMM, 8 no expand MFMA,
# 16 no weight DMA, 32 no store, 64 no X loads, 128 no chunk loop; 63 = every phase off = the skeleton).
# Run on the GPU box:  ABL_LIST="0 2 4 8 63" bash tools/abl.sh
cd /tmp && export TMPDIR=/tmp
for dbg in ${ABL_LIST:-0 2 4 8 16 63 127 191 255}; do
  export BIRDA_HIP_MB_DBG=$dbg
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/abl_$dbg -- python3 $GRAFT_REPO_ROOT/tools/gpu_quick_bench.py birdnet_v24 1000 1000 > /dev/null 2>&1
  echo "== dbg $dbg"; python3 $GRAFT_REPO_ROOT/tools/kstats.py $GRAFT_REPO_ROOT/gpurun_out/abl_$dbg | grep mbconv | sort | awk '{print $1, $5}' | tr '\n' ' '; echo
done
