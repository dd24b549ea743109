#!/bin/bash
# per-kernel time table of one bench configuration: bash tools/kstats.sh <tag> [bench args]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-ks}; shift
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o ks -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end --no-configs "$@" > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
python3 - $O <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    n = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
    print(f"{n:62s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:9.1f} us  {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
