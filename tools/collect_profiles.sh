#!/bin/bash
# gpurun_out/refresh (tools/refresh_profiles.sh) -> profiles/r${ROUND}_*: one set per round.
ROUND=${ROUND:-06}
R=gpurun_out/refresh; P=profiles
cp $R/bench_c2.json $P/r${ROUND}_bench.json; cp $R/bench_c2_depth1.json $P/r${ROUND}_bench_depth1.json; cp $R/bench_c2_under_rocprof.json $P/r${ROUND}_bench_under_rocprof.json
cp $R/perop_c2.txt $P/r${ROUND}_perop.txt; cp $R/kernel_stats_c2.txt $P/r${ROUND}_kernel_stats.txt; cp $R/kernel_stats_c2_in_flight.txt $P/r${ROUND}_kernel_stats_in_flight.txt
cp $R/pmc_fetch_c2.txt $P/r${ROUND}_pmc_fetch.txt; cp $R/pmc_write_c2.txt $P/r${ROUND}_pmc_write.txt; cp $R/pmc_sq_c2.txt $P/r${ROUND}_pmc_sq.txt; cp $R/pmc_l2_c2.txt $P/r${ROUND}_pmc_l2.txt
cp $R/traffic.json $P/r${ROUND}_traffic.json; cp $R/tuned_c2.json $P/r${ROUND}_tuned.json
cp $R/bench_c2_force_dist.json $P/r${ROUND}_bench_force_dist_1rank_rccl.json; cp $R/bench_c2_force_dist_depth1.json $P/r${ROUND}_bench_force_dist_1rank_rccl_depth1.json
for c in "c3 c3_effb0lite_bf16 _efficientnetb0lite_416_b128_bf16" "c4 c4_mbv2x14_f32 _mobilenetv2x14_512_b64_f32" "c5 c5_effb3lite_f16 _efficientnetb3lite_640_b32_f16" "c3se effb0_bf16 _efficientnetb0_416_b128_bf16" "c5se effb3_f16 _efficientnetb3_640_b32_f16"; do set -- $c
  cp $R/bench_$1.json $P/r${ROUND}_bench_$2.json; cp $R/bench_$1_depth1.json $P/r${ROUND}_bench_$2_depth1.json; cp $R/bench_$1_under_rocprof.json $P/r${ROUND}_bench_$2_under_rocprof.json
  cp $R/perop_$1.txt $P/r${ROUND}_perop_$2.txt; cp $R/kernel_stats_$1.txt $P/r${ROUND}_kernel_stats_$2.txt; cp $R/kernel_stats_$1_in_flight.txt $P/r${ROUND}_kernel_stats_$2_in_flight.txt
  cp $R/pmc_fetch_$1.txt $P/r${ROUND}_pmc_fetch_$2.txt; cp $R/pmc_write_$1.txt $P/r${ROUND}_pmc_write_$2.txt
  [ -f $R/pmc_sq_$1.txt ] && cp $R/pmc_sq_$1.txt $P/r${ROUND}_pmc_sq_$2.txt; [ -f $R/pmc_l2_$1.txt ] && cp $R/pmc_l2_$1.txt $P/r${ROUND}_pmc_l2_$2.txt
  cp $R/traffic$3.json $P/r${ROUND}_traffic$3.json
done
cp $R/bench_c2_bf16.json $P/r${ROUND}_bench_c2_bf16.json
cp $R/summary.txt $P/r${ROUND}_configs_summary.txt
python -m yoloret_amd.build --report > $P/r${ROUND}_kernel_regs.txt
ls $P | grep -c "^r${ROUND}_"
