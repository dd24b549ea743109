#!/bin/bash
# One-call measurement sweep on the GPU box (run from the repo root via gpurun).  Writes under
# gpurun_out/r01/; the summaries judged are copied to profiles/ afterwards.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r01
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/headline_bench.json 2> $O/headline_bench.err && tail -c 600 $O/headline_bench.json && echo &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_headline -o headline -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/prof_headline.log 2>&1 &&
python3 $R/bench.py --workload cfg3-sage-512x360-h128 > $O/cfg3_bench.json 2> $O/cfg3_bench.err && tail -c 400 $O/cfg3_bench.json && echo &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg3 -o cfg3 -- python3 $R/bench.py --workload cfg3-sage-512x360-h128 --steps 10 --warmup 3 --no-cpu-baseline > $O/prof_cfg3.log 2>&1 &&
python3 $R/bench.py --workload cfg2-gcn-512x84-h64 > $O/cfg2_bench.json 2> $O/cfg2_bench.err &&
python3 $R/bench.py --workload cfg2-gcn-512x84-h64 --graph --no-cpu-baseline > $O/cfg2_graph_bench.json 2> $O/cfg2_graph_bench.err &&
python3 $R/bench.py --workload cfg5-gcn-64x1000-h256-fp32 --steps 10 --warmup 3 > $O/cfg5_fp32_bench.json 2> $O/cfg5_fp32_bench.err &&
python3 $R/tools/scatter_bench.py > $O/scatter_bench.log 2>&1 && cp $R/gpurun_out/scatter_bench.json $O/ &&
python3 $R/tools/gemm_bench.py > $O/gemm_bench.log 2>&1 && cp $R/gpurun_out/gemm_bench.json $O/ &&
for wl in cfg4-headline-gcn-4096x360-h64 cfg3-sage-512x360-h128; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_${wl}_f -o f -- python3 $R/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_${wl}_f.log 2>&1 &&
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_${wl}_w -o w -- python3 $R/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_${wl}_w.log 2>&1 &&
  python3 $R/tools/pmc_summarise.py $O/pmc_${wl}_f/f_counter_collection.csv $O/pmc_${wl}_w/w_counter_collection.csv $wl 4 $O/pmc_${wl}.json || exit 1
done
echo SWEEP-OK
