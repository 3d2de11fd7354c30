# per-kernel time of the fused blocks under different BIRDA_HIP_MB_PREFER lists: bash tools/cmp_cfg.sh "48" "61" ...
cd /tmp && export TMPDIR=/tmp; export BIRDA_HIP_PRECISION=${BIRDA_HIP_PRECISION:-f16x3}
i=0
for pref in "$@"; do
  i=$((i+1)); export BIRDA_HIP_MB_PREFER=$pref
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/cmpcfg_$i -- python3 $GRAFT_REPO_ROOT/tools/gpu_quick_bench.py birdnet_v24 1000 1000 > /dev/null 2>&1
  echo "== prefer $pref"; python3 $GRAFT_REPO_ROOT/tools/kstats.py $GRAFT_REPO_ROOT/gpurun_out/cmpcfg_$i | grep mbconv | sort | awk '{printf "%s %s | ", $1, $5}'; echo
done
