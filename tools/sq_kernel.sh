#!/bin/bash
# SQ issue/wait breakdown of the kernels of any python script (diagnostic; separate --pmc passes).
# usage (via gpurun, repo root): bash tools/sq_kernel.sh <tag> <kernel-name-prefix> <script.py> [args...]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; PREFIX=$2; shift 2
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES SQ_INSTS_VALU"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
P3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAIT_INST_ANY"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/p$i -o p -- python3 "$@" > $O/p$i.log 2>&1 || { tail -5 $O/p$i.log; exit 1; }
done
python3 - "$O" "$PREFIX" <<'PY'
import csv, sys, collections, glob, json
O, prefix = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, d in acc.items():
    if not k.startswith(prefix):
        continue
    out[k] = {c: sum(v) / len(v) for c, v in d.items()}
json.dump(out, open(O + "/sq_summary.json", "w"), indent=1)
for k, d in out.items():
    wc = d.get("SQ_WAVE_CYCLES", 1)
    print(k)
    for c in sorted(d):
        print(f"   {c:28s} {d[c]:16.0f}  {d[c] / wc:8.3f} of WAVE_CYCLES")
PY
