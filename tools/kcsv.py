#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats CSV compactly: short kernel name, calls, average us, total us, %."""
import csv, re, sys
for path in sys.argv[1:]:
    print("==", path)
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        n = r["Name"]
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"^void ", "", n)
        n = n.split("(")[0][:60]
        print(f"{n:60s} {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:9.1f} {float(r['TotalDurationNs'])/1e3:10.1f} {float(r['Percentage']):6.2f}")
