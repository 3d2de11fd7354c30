cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
cp birda_amd/libbirda_hip.so /tmp/new.so
{
python -m pytest tests/test_parity_gpu.py -x -q -k "v30 or squeeze or perch or wide" 2>&1 | tail -3
for g in old new; do [ $g = old ] && cp tools/ab/libbirda_hip_old.so birda_amd/libbirda_hip.so || cp /tmp/new.so birda_amd/libbirda_hip.so; echo "v30 $g"; python tools/gpu_layer_times.py birdnet_v30_sized 256 > /tmp/lt_$g.txt 2>&1; python - <<PY
import re
t=0
for l in open("/tmp/lt_$g.txt"):
    m=re.match(r"layer\s+(\d+)\s+(.*?)\s+([\d.]+) us per", l)
    if m: t+=float(m.group(3))
print("sum of layers us per 256:", round(t,1))
PY
grep -E "layer 5(59|60|61|62) " /tmp/lt_$g.txt
done
cp /tmp/new.so birda_amd/libbirda_hip.so
} > gpurun_out/lp_gate.txt 2>&1
