#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as the guide prescribes)
into HBM bytes per launch of every library kernel.
usage: pmc_summarise.py <fetch_counter_collection.csv> <write_counter_collection.csv> <workload> <steps+warmup> <out.json>"""
import collections
import csv
import json
import sys

fetch_csv, write_csv, workload, nsteps, out_path = sys.argv[1:6]
nsteps = int(nsteps)
SETUP = {"k_zero_i32", "k_bell_width", "k_scan_tile_sums", "k_scan_top", "k_scan_apply", "k_bell_fill",
         "k_csr_grouped", "k_gather_f32", "k_count", "k_fill", "k_sort_rows"}


def load(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def short(k):
    return k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]


f, w = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
doc = {"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python3 bench.py "
                 f"--workload {workload} (steps + warm-up = {nsteps}) --no-cpu-baseline",
       "workload": workload,
       "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE tallies 64 B per 128-B request "
                     "of a wide coalesced stream (MI355X_MICROARCH.md, HBM), WRITE_SIZE is exact; units are KiB",
       "kernels": {}}
total = 0.0
for k in f:
    s = short(k)
    if s in SETUP or s.startswith("at::") or s.startswith("__amd") or "Cijk" in s:
        continue
    fa = sum(f[k]) / len(f[k])
    wl = w.get(k, [0.0])
    wa = sum(wl) / len(wl)
    b = (2 * fa + wa) * 1024
    doc["kernels"][s] = {"FETCH_SIZE_KiB_avg": fa, "WRITE_SIZE_KiB_avg": wa, "launches": len(f[k]),
                         "hbm_bytes_per_launch": b}
    total += b * len(f[k]) / nsteps
doc["hbm_bytes_per_step_library_kernels"] = total
json.dump(doc, open(out_path, "w"), indent=1)
print(workload, "HBM bytes per step (library kernels):", round(total / 1e9, 3), "GB")
