cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/long
FUZZ_PROCESS=1 timeout 1200 python tools/fuzz_wav_decoder.py 6000 101 > gpurun_out/long/wav.txt 2>&1
timeout 1200 python tools/fuzz_create.py 6000 102 > gpurun_out/long/create.txt 2>&1
timeout 1500 python tools/soak_random_plans.py 1000 1700 > gpurun_out/long/plans.txt 2>&1
timeout 600 python tools/soak_random_plans.py 2070 2130 >> gpurun_out/long/plans.txt 2>&1
