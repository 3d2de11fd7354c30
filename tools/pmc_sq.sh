#!/bin/bash
# SQ counter pass (own run, kernel-trace only): where the waves' cycles go per kernel
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/${1:-sq}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export BIRDA_HIP_PRECISION=${BIRDA_HIP_PRECISION:-f16x3}   # the mode bench.py runs (exported here: no env hop under rocprofv3)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $out/p1 -- python3 $root/tools/gpu_quick_bench.py birdnet_v24 1000 1000 > /dev/null 2> $out/p1.log
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/p2 -- python3 $root/tools/gpu_quick_bench.py birdnet_v24 1000 1000 > /dev/null 2> $out/p2.log
python3 - <<PY
import csv, glob, re
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int); dur = defaultdict(float)
for sub in ("p1", "p2"):
    for path in glob.glob("$out/%s/**/*counter_collection.csv" % sub, recursive=True):
        seen = set()
        for r in csv.DictReader(open(path)):
            n = r["Kernel_Name"]; m = re.search(r"mbconv_kernel<([^>]*)>", n)
            name = "mbconv<" + m.group(1).replace(" ", "") + ">" if m else re.sub(r"\(.*", "", n).replace("void ", "")
            acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
            if sub == "p1" and (r["Dispatch_Id"]) not in seen:
                seen.add(r["Dispatch_Id"]); cnt[name] += 1; dur[name] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
keys = ["SQ_WAVE_CYCLES","SQ_BUSY_CYCLES","SQ_VALU_MFMA_BUSY_CYCLES","SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_LDS","SQ_WAIT_INST_LDS","SQ_LDS_BANK_CONFLICT","SQ_LDS_IDX_ACTIVE","SQ_INSTS_VALU","SQ_INSTS_MFMA","GRBM_GUI_ACTIVE"]
for name in sorted(acc, key=lambda k: -dur[k])[:14]:
    a = acc[name]; wc = a["SQ_WAVE_CYCLES"] or 1
    print(f"{name[:46]:46s} n={cnt[name]:2d} us={dur[name]/max(cnt[name],1)/1e3:8.1f} " + " ".join(f"{k.replace('SQ_','')[:14]}={a[k]/wc:6.3f}" for k in keys[2:11]) + f" valu/mfma_insts={a['SQ_INSTS_VALU']/max(a['SQ_INSTS_MFMA'],1):5.2f} mfma_busy/busy={a['SQ_VALU_MFMA_BUSY_CYCLES']/max(a['SQ_BUSY_CYCLES'],1):6.3f}")
PY
