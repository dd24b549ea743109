#!/bin/bash
# A/B of library builds on ONE box through bench.py + rocprofv3 kernel stats:
#   tools/ab_bench.sh "<bench args>" "<kernel name regex>" "<-D flags A>" "<-D flags B>" ...
cd "$(dirname "$0")/.."
R=$(pwd)
ARGS=$1; PAT=$2; shift 2
i=0
for flags in "$@"; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function -Wno-pass-failed $flags -shared \
    -o /tmp/libcgnn_ab_$i.so connectome_gnn_amd/csrc/*.hip || exit 1
  i=$((i+1))
done
cd /tmp && export TMPDIR=/tmp
i=0
for flags in "$@"; do
  echo "== [$flags]"
  rm -rf /tmp/abp_$i
  CGNN_LIB=/tmp/libcgnn_ab_$i.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abp_$i -o ks -- python3 $R/bench.py $ARGS --steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end > /tmp/abp_$i.log 2>&1 || { tail -3 /tmp/abp_$i.log; exit 1; }
  grep -o '"ms_per_step": [0-9.]*' /tmp/abp_$i.log
  python3 - /tmp/abp_$i "$PAT" <<'PY'
import csv, sys, glob, re
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
    if re.search(sys.argv[2], n):
        print(f"   {n:62s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
  i=$((i+1))
done
