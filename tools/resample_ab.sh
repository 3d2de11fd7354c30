cd $GRAFT_REPO_ROOT
timeout 1200 python3 -m pytest tests/test_resampler_gpu.py -x -q -m gpu 2>&1 | tail -4
for f in 0 1; do echo "== BIRDA_HIP_RESAMPLE_F32=$f"; BIRDA_HIP_RESAMPLE_F32=$f python3 tools/gpu_extra_bench.py 2>&1 | grep -E "resample"; done
