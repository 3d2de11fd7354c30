"""Print a rocprofv3 kernel_stats.csv compactly (mbconv kernels by template arguments).  With the kernel trace beside it
(*kernel_trace.csv of the same run) a last column gives the average over each kernel's LAST `tail` calls (argv[2], default 50):
bench.py's pre-warm and warm-up launches run on a cooler, faster chip than the timed regions, so the average over all calls
undercuts what bench.py's HIP events see in its timed region by ~4 %; the tail is that region."""
import csv, glob, re, sys
from collections import defaultdict
root = sys.argv[1]
tail = int(sys.argv[2]) if len(sys.argv) > 2 else 50


def short(n):
    m = re.search(r"mbconv_kernel<([^>]*)>", n)
    return "mbconv<" + m.group(1).replace(" ", "") + ">" if m else re.sub(r"\(.*", "", n)[:50]


last = {}
traces = sorted(glob.glob(root + "/**/*kernel_trace.csv", recursive=True))
if traces:
    calls = defaultdict(list)
    for r in csv.DictReader(open(traces[-1])):
        calls[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    for k, v in calls.items():
        v.sort()
        d = [x[1] for x in v[-tail:]]
        last[short(k)] = sum(d) / len(d) / 1e3
path = sorted(glob.glob(root + "/**/*kernel_stats.csv", recursive=True))[-1]
for r in csv.DictReader(open(path)):
    name = short(r["Name"])
    extra = f"  last {tail}: {last[name]:8.1f}" if name in last else ""
    print(f"{name:52s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:8.2f} {float(r['Percentage']):5.1f}%{extra}")
