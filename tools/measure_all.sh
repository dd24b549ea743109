#!/bin/bash
# One-call measurement sweep on the GPU box (run from the repo root via gpurun).  Writes under
# gpurun_out/$TAG/ (TAG = r04 by default); tools/collect_profiles.sh copies the judged summaries
# to profiles/ afterwards.  Kernel-trace and --pmc passes are separate runs (FETCH_SIZE and
# WRITE_SIZE in passes of their own), as MI355X_MICROARCH.md prescribes.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${TAG:-r04}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
H=cfg4-headline-gcn-4096x360-h64
ONE="--no-configs --no-cpu-baseline --no-end-to-end"
# 1. HBM traffic: FETCH_SIZE / WRITE_SIZE passes of every workload, summarised into profiles/ OF THIS COPY so that
#    the bench line of step 2 reads fresh figures (tools/collect_profiles.sh copies them home afterwards)
export CGNN_BENCH_COLLECTING_PMC=1
for spec in "$H - headline" "cfg3-sage-512x360-h128 - cfg3" "cfg2-gcn-512x84-h64 - cfg2" "cfg5-gcn-64x1000-h256-fp16 - cfg5_fp16" "cfg5-gcn-64x1000-h256-fp32 - cfg5_fp32" "shard512-gcn-512x360-h64 512 shard512"; do
  set -- $spec
  wl=$1; extra=""; bwl=$wl
  if [ "$2" != "-" ]; then extra="--batch $2"; bwl=$H; fi
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_${wl}_f -o f -- python3 $R/bench.py --workload $bwl $extra --launch eager --steps 3 --warmup 1 $ONE > $O/pmc_${wl}_f.log 2>&1 &&
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_${wl}_w -o w -- python3 $R/bench.py --workload $bwl $extra --launch eager --steps 3 --warmup 1 $ONE > $O/pmc_${wl}_w.log 2>&1 &&
  python3 $R/tools/pmc_summarise.py $O/pmc_${wl}_f/f_counter_collection.csv $O/pmc_${wl}_w/w_counter_collection.csv $wl 4 $O/pmc_${wl}.json || exit 1
  cp $O/pmc_${wl}.json $R/profiles/${TAG}_$3_pmc_traffic.json
  echo "pmc $wl ok"
done
unset CGNN_BENCH_COLLECTING_PMC
# 2. the driver's command: headline + every single-GPU config in one JSON line
python3 $R/bench.py > $O/headline_bench.json 2> $O/headline_bench.err && tail -c 600 $O/headline_bench.json && echo &&
# 3. kernel traces
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_headline -o headline -- python3 $R/bench.py --steps 10 --warmup 3 $ONE > $O/prof_headline.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_shard512 -o ks -- python3 $R/bench.py --batch 512 --launch eager --steps 10 --warmup 3 $ONE > $O/prof_shard512.log 2>&1 &&
for wl in cfg3-sage-512x360-h128 cfg2-gcn-512x84-h64 cfg5-gcn-64x1000-h256-fp16 cfg5-gcn-64x1000-h256-fp32; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$wl -o ks -- python3 $R/bench.py --workload $wl --launch eager --steps 10 --warmup 3 $ONE > $O/prof_$wl.log 2>&1 || exit 1
  echo "kernel trace $wl ok"
done
python3 $R/tools/scatter_bench.py > $O/scatter_bench.log 2>&1 && cp $R/gpurun_out/scatter_bench.json $O/ &&
echo SWEEP-OK
