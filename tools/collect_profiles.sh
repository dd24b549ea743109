#!/bin/bash
# copy the judged summaries of a tools/measure_all.sh sweep (gpurun_out/r02/) into profiles/r02_*
cd "$(dirname "$0")/.."
S=gpurun_out/r02; P=profiles
cp $S/headline_bench.json $P/r02_headline_bench.json
cp $S/headline_shard512_graph_bench.json $P/r02_headline_shard512_graph_bench.json
cp $S/prof_headline/headline_kernel_stats.csv $P/r02_headline_kernel_stats.csv
cp $S/pmc_cfg4-headline-gcn-4096x360-h64.json $P/r02_headline_pmc_traffic.json
for pair in "cfg2-gcn-512x84-h64 cfg2_gcn_h64 cfg2" "cfg3-sage-512x360-h128 cfg3_sage_h128 cfg3" \
            "cfg5-gcn-64x1000-h256-fp16 cfg5_gcn_fp16 cfg5_fp16" "cfg5-gcn-64x1000-h256-fp32 cfg5_gcn_fp32 -"; do
  set -- $pair
  cp $S/$1_eager_bench.json $P/r02_$2_eager_bench.json
  cp $S/$1_graph_bench.json $P/r02_$2_graph_bench.json
  cp $S/prof_$1/ks_kernel_stats.csv $P/r02_$2_kernel_stats.csv
  [ "$3" != "-" ] && cp $S/pmc_$1.json $P/r02_$3_pmc_traffic.json
done
cp $S/scatter_bench.json $P/r02_cfg5_scatter_bench.json
ls -la $P | grep r02 | wc -l
