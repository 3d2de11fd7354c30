#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE passes) of the default bench under two settings of an environment switch:
#   bash tools/pmc_ab.sh TAG VAR val1 val2      (the variable is exported before rocprofv3: no env wrapper after "--")
tag=$1; var=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  out=$root/gpurun_out/${tag}_${var}_$v; mkdir -p $out
  export $var=$v
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > /dev/null 2> $out/pmc_fetch.log
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > /dev/null 2> $out/pmc_write.log
  python3 $root/tools/pmc_summary.py $out $out/traffic.json > $out/traffic.txt
  echo "== $var=$v"; head -4 $out/traffic.txt
  rm -rf $out/pmc_fetch $out/pmc_write
done
