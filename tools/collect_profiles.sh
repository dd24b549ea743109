#!/bin/bash
# copy the judged summaries of a tools/measure_all.sh sweep (gpurun_out/$TAG/) into profiles/$TAG_*
cd "$(dirname "$0")/.."
TAG=${TAG:-r04}
S=gpurun_out/$TAG; P=profiles
cp $S/headline_bench.json $P/${TAG}_bench_all_configs.json
cp $S/prof_headline/headline_kernel_stats.csv $P/${TAG}_headline_kernel_stats.csv
cp $S/prof_shard512/ks_kernel_stats.csv $P/${TAG}_shard512_kernel_stats.csv
cp $S/pmc_cfg4-headline-gcn-4096x360-h64.json $P/${TAG}_headline_pmc_traffic.json
cp $S/pmc_shard512-gcn-512x360-h64.json $P/${TAG}_shard512_pmc_traffic.json
for pair in "cfg2-gcn-512x84-h64 cfg2_gcn_h64 cfg2" "cfg3-sage-512x360-h128 cfg3_sage_h128 cfg3" \
            "cfg5-gcn-64x1000-h256-fp16 cfg5_gcn_fp16 cfg5_fp16" "cfg5-gcn-64x1000-h256-fp32 cfg5_gcn_fp32 cfg5_fp32"; do
  set -- $pair
  cp $S/prof_$1/ks_kernel_stats.csv $P/${TAG}_$2_kernel_stats.csv
  [ "$3" != "-" ] && cp $S/pmc_$1.json $P/${TAG}_$3_pmc_traffic.json
done
cp $S/scatter_bench.json $P/${TAG}_cfg5_scatter_bench.json
ls $P | grep ${TAG}_ | wc -l
