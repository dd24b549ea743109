#!/usr/bin/env python3
"""Idle time before each kernel of the last traced step: python tools/gaps.py <kernel_trace.csv> <name part>
(rows where the kernel or its predecessor matches)."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "").split("(")[0][:44] for r in rows]
idx = [i for i, n in enumerate(names) if n.startswith("k_adam")]
i1, i0 = idx[-1], idx[-2]
for k in range(i0 + 1, i1 + 1):
    r = rows[k]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    gap = (int(r["Start_Timestamp"]) - int(rows[k - 1]["End_Timestamp"])) / 1e3
    if pat in names[k] or pat in names[k - 1]:
        print(f"{names[k]:46s} {d:7.1f} us  gap {gap:6.1f} lds {r.get('LDS_Block_Size', '')}")
