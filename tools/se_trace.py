"""Per-call durations of the squeeze-excite gate kernels in the LAST forward of a rocprofv3 --kernel-trace run (call order = block order)."""
import csv, glob, sys, re
root=sys.argv[1]
f=sorted(glob.glob(root+"/**/*kernel_trace.csv", recursive=True))[-1]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
last=max(i for i,r in enumerate(rows) if "topk" in r["Kernel_Name"])
prev=max(i for i,r in enumerate(rows[:last]) if "topk" in r["Kernel_Name"])
tot=0
for r in rows[prev+1:last+1]:
    n=r["Kernel_Name"]
    if "se_" in n:
        d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3; tot+=d
        print(re.sub(r"\(.*","",n)[:30], "grid", int(r["Grid_Size_X"])//256, r["Grid_Size_Y"], f"{d:7.1f} us")
fw=(int(rows[last]["End_Timestamp"])-int(rows[prev]["End_Timestamp"]))/1e3
print(f"gate kernels {tot:.0f} us of a forward of {fw:.0f} us")
