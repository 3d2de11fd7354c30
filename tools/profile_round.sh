#!/bin/bash
# Round profile: kernel-trace stats of the default bench command + two PMC passes (HBM bytes).
# Run on the GPU box:  bash tools/profile_round.sh r1_b
tag=${1:-r1}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra-legs > $out/bench_under_rocprof.json 2> $out/trace.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > /dev/null 2> $out/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > /dev/null 2> $out/pmc_write.log
python3 $root/bench.py --steps 20 --warmup 3 > $out/bench.json 2> $out/bench.log
python3 $root/tools/kstats.py $out/trace > $out/kernel_stats.txt
python3 $root/tools/pmc_summary.py $out $out/traffic.json > $out/traffic.txt
ls -R $out | head -40
