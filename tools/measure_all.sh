#!/bin/bash
# One-call measurement sweep on the GPU box (run from the repo root via gpurun).  Writes under
# gpurun_out/r02/; the summaries judged are copied to profiles/ afterwards.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
H=cfg4-headline-gcn-4096x360-h64
python3 $R/bench.py > $O/headline_bench.json 2> $O/headline_bench.err && tail -c 900 $O/headline_bench.json && echo &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_headline -o headline -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end > $O/prof_headline.log 2>&1 &&
python3 $R/bench.py --batch 512 --launch graph --no-cpu-baseline --no-end-to-end > $O/headline_shard512_graph_bench.json 2> $O/shard512.err &&
for wl in cfg3-sage-512x360-h128 cfg2-gcn-512x84-h64 cfg5-gcn-64x1000-h256-fp16 cfg5-gcn-64x1000-h256-fp32; do
  python3 $R/bench.py --workload $wl --launch eager > $O/${wl}_eager_bench.json 2> $O/${wl}_eager.err &&
  python3 $R/bench.py --workload $wl --launch graph --no-cpu-baseline > $O/${wl}_graph_bench.json 2> $O/${wl}_graph.err &&
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$wl -o ks -- python3 $R/bench.py --workload $wl --launch eager --steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end > $O/prof_$wl.log 2>&1 || exit 1
done
python3 $R/tools/scatter_bench.py > $O/scatter_bench.log 2>&1 && cp $R/gpurun_out/scatter_bench.json $O/ &&
for wl in $H cfg3-sage-512x360-h128 cfg2-gcn-512x84-h64 cfg5-gcn-64x1000-h256-fp16; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_${wl}_f -o f -- python3 $R/bench.py --workload $wl --launch eager --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/pmc_${wl}_f.log 2>&1 &&
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_${wl}_w -o w -- python3 $R/bench.py --workload $wl --launch eager --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/pmc_${wl}_w.log 2>&1 &&
  python3 $R/tools/pmc_summarise.py $O/pmc_${wl}_f/f_counter_collection.csv $O/pmc_${wl}_w/w_counter_collection.csv $wl 4 $O/pmc_${wl}.json || exit 1
done
echo SWEEP-OK
