"""Print a rocprofv3 kernel_stats.csv compactly (mbconv kernels by template arguments)."""
import csv, glob, re, sys
path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[-1]
for r in csv.DictReader(open(path)):
    n = r["Name"]
    m = re.search(r"mbconv_kernel<([^>]*)>", n)
    name = "mbconv<" + m.group(1).replace(" ", "") + ">" if m else re.sub(r"\(.*", "", n)[:50]
    print(f"{name:52s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:8.2f} {float(r['Percentage']):5.1f}%")
