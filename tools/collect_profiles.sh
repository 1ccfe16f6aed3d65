#!/bin/bash
# Copy a round's judged summaries from gpurun_out/$R (scratch) into profiles/ (tracked).  Run in the authoring container
# after `gpurun -- bash tools/profile_round.sh`.
R=${EGOEGO_ROUND:-r06}
O=gpurun_out/$R
set -e
for f in bench_b256_t120 bench_b32_t120 bench_b64_t120 bench_b128_t120 bench_b256_t196 bench_2ranks_gloo_one_gpu bench_1rank_rccl_forced_gather bench_b256_t120_trained_like bench_b256_t120_trained_like_p3 bench_b256_t196_trained_like bench_b64_t120_trained_like; do
  grep -h "^{" $O/$f.json | tail -1 > profiles/${R}_$f.json
done
cp $O/step_times.jsonl profiles/${R}_step_times.jsonl
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) profiles/${R}_bench_b256_t120_kernel_stats.csv
cp $(find $O/stats_b32 -name "*kernel_stats.csv" | head -1) profiles/${R}_bench_b32_t120_kernel_stats.csv
cp $(find $O/stats_t196 -name "*kernel_stats.csv" | head -1) profiles/${R}_bench_b256_t196_kernel_stats.csv
if [ -d $O/stats_p3 ]; then
  cp $(find $O/stats_p3 -name "*kernel_stats.csv" | head -1) profiles/${R}_bench_b256_t120_p3_kernel_stats.csv
  cp $(find $O/stats_p3_t196 -name "*kernel_stats.csv" | head -1) profiles/${R}_bench_b256_t196_p3_kernel_stats.csv
  cp $O/step_times_p3.jsonl profiles/${R}_step_times_p3.jsonl
fi
A=$(find $O/pmc_a -name "*counter_collection.csv" | head -1); B=$(find $O/pmc_b -name "*counter_collection.csv" | head -1); C=$(find $O/pmc_c -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py $A $B $C > profiles/${R}_pmc_per_kernel.csv
python3 tools/fetch_calib.py $(find $O/calib_f -name "*counter_collection.csv" | head -1) $(find $O/calib_w -name "*counter_collection.csv" | head -1) > profiles/${R}_fetch_calibration.json
EGOEGO_ROUND=$R python3 tools/traffic_json.py $A $B 256 120 9
if [ -d $O/pmc_a_p3 ]; then
  A3=$(find $O/pmc_a_p3 -name "*counter_collection.csv" | head -1); B3=$(find $O/pmc_b_p3 -name "*counter_collection.csv" | head -1); C3=$(find $O/pmc_c_p3 -name "*counter_collection.csv" | head -1)
  python3 tools/pmc_summary.py $A3 $B3 $C3 > profiles/${R}_pmc_per_kernel_p3.csv
  EGOEGO_ROUND=$R python3 tools/traffic_json.py $A3 $B3 256 120 3 _p3
fi
EGOEGO_ROUND=$R python3 tools/roofline_report.py
