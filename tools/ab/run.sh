cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
cp birda_amd/libbirda_hip.so /tmp/new.so
use() { cp tools/ab/libbirda_hip_$1.so birda_amd/libbirda_hip.so; }
{
for cfg in "1000 576 24 1" "1000 816 32 1" "1000 1392 56 1" "1000 2304 96 1" "256 2304 96 1" "256 40 8 128" "256 144 4 64" "256 192 8 32" "256 288 12 8"; do tools/microbench/se_gate.bin $cfg; done
for g in head g577 g250 g150; do use $g; echo "== $g"
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tr && rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 $GRAFT_REPO_ROOT/tools/gpu_quick_bench.py perch_v2 256 256 > /dev/null 2>&1; python3 $GRAFT_REPO_ROOT/tools/se_trace.py /tmp/tr )
done
for r in 1 2; do for g in head g577 g250 g150; do use $g; for mb in 256 1000; do echo -n "c4 $g mb $mb "; timeout 300 python bench.py --config c4 --micro-batch $mb --no-cpu-baseline --no-extra-legs --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['config'].get('sclk_mhz'), d['config'].get('power_w'))"
done; done; done
use g250
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "perch or squeeze or v30 or gate or se_" 2>&1 | tail -2
cp /tmp/new.so birda_amd/libbirda_hip.so
} > gpurun_out/gate_ab.txt 2>&1
