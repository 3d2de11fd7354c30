cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/zz
python bench.py --gpus 1 --steps 20 --warmup 3 > gpurun_out/zz/bench.json 2> gpurun_out/zz/bench.log
python bench.py --config c4 --steps 10 --warmup 2 > gpurun_out/zz/bench_c4.json 2>> gpurun_out/zz/bench.log
python bench.py --config c4 --micro-batch 256 --steps 10 --warmup 2 --no-cpu-baseline --no-extra-legs > gpurun_out/zz/bench_c4_mb256.json 2>> gpurun_out/zz/bench.log
python tools/gpu_latency.py > gpurun_out/zz/latency.txt 2>&1
python tools/gpu_latency_ll.py >> gpurun_out/zz/latency.txt 2>&1
