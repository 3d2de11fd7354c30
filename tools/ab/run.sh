cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
cp birda_amd/libbirda_hip.so /tmp/new.so
use() { [ $1 = new ] && cp /tmp/new.so birda_amd/libbirda_hip.so || cp tools/ab/libbirda_hip_$1.so birda_amd/libbirda_hip.so; }
{
for r in 1 2 3; do for g in old new; do use $g; for cfg in c2 c4; do for mb in 256 512; do echo -n "$cfg $g mb $mb "; timeout 300 python bench.py --config $cfg --micro-batch $mb --no-cpu-baseline --no-extra-legs --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['config'].get('sclk_mhz'), d['config'].get('power_w'))"
done; done; done; done
use new
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q 2>&1 | tail -2
} > gpurun_out/ntb_ab.txt 2>&1
