cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
{
REPS=3 bash tools/ab.sh lib
python -m pytest tests/test_parity_gpu.py tests/test_random_plans_gpu.py -x -q 2>&1 | tail -2
for sd in 309 451 1001; do LIBX=1 bash tools/ab.sh x python tools/gpu_f32_pad_rule.py $sd 256 2>&1 | grep seed; done
} > gpurun_out/ab_lib.txt 2>&1
