#!/bin/bash
# Measurement artifacts of a round (run on the GPU box: gpurun -- 'bash tools/profile_round.sh').  Everything lands under
# gpurun_out/$R/; the summaries that are judged are copied into profiles/ afterwards (tools/collect_profiles.sh).
#   1. the bench lines                                   python bench.py [--batch ..] [--window ..]
#   2. per-kernel time summaries                         rocprofv3 --kernel-trace --stats  (same commands, fewer steps)
#   3. PMC counters, three separate passes               rocprofv3 --kernel-trace --pmc ...   (no --sys-trace etc.)
# (the rocprofv3 runs pass --precision 9 — what "auto" picks for the synthetic weights — so that the probe of "auto" does not mix its small-batch launches into the per-kernel averages)
R=${EGOEGO_ROUND:-r06}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$R; mkdir -p $O
python3 bench.py --steps 200 --warmup 10 > $O/bench_b256_t120.json 2> $O/bench.err
for b in 32 64 128; do python3 bench.py --steps 200 --warmup 10 --batch $b --no-cpu-baseline > $O/bench_b${b}_t120.json 2>> $O/bench.err; done
python3 bench.py --steps 100 --warmup 10 --batch 256 --window 196 --no-cpu-baseline > $O/bench_b256_t196.json 2>> $O/bench.err
EGOEGO_DIST_BACKEND=gloo python3 bench.py --gpus 2 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_2ranks_gloo_one_gpu.json 2>> $O/bench.err
EGOEGO_FORCE_COLLECTIVE=1 python3 bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_1rank_rccl_forced_gather.json 2>> $O/bench.err
python3 bench.py --steps 200 --warmup 10 --weights trained-like --no-cpu-baseline > $O/bench_b256_t120_trained_like.json 2>> $O/bench.err
python3 bench.py --steps 200 --warmup 10 --weights trained-like --precision 3 --no-cpu-baseline > $O/bench_b256_t120_trained_like_p3.json 2>> $O/bench.err
python3 bench.py --steps 100 --warmup 10 --window 196 --weights trained-like --no-cpu-baseline > $O/bench_b256_t196_trained_like.json 2>> $O/bench.err
python3 bench.py --steps 200 --warmup 10 --batch 64 --weights trained-like --no-cpu-baseline > $O/bench_b64_t120_trained_like.json 2>> $O/bench.err
python3 tools/step_times.py --steps 100 --batches 1,16,24,32,64,128,256 --windows 120,196 --api > $O/step_times.jsonl 2>> $O/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o stats -- python3 bench.py --steps 20 --warmup 3 --precision 9 --no-probe --no-cpu-baseline > $O/stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_b32 -o stats -- python3 bench.py --steps 50 --warmup 3 --batch 32 --precision 9 --no-probe --no-cpu-baseline > $O/stats_b32.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_t196 -o stats -- python3 bench.py --steps 20 --warmup 3 --window 196 --precision 9 --no-probe --no-cpu-baseline > $O/stats_t196.log 2>&1
# split-bf16 (what "auto" runs on a checkpoint whose chain amplifies operand rounding): per-kernel summaries and small-batch step times
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_p3 -o stats -- python3 bench.py --steps 20 --warmup 3 --precision 3 --no-probe --no-cpu-baseline > $O/stats_p3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_p3_t196 -o stats -- python3 bench.py --steps 20 --warmup 3 --window 196 --precision 3 --no-probe --no-cpu-baseline > $O/stats_p3_t196.log 2>&1
python3 tools/step_times.py --steps 100 --batches 1,2,8,32 --windows 120,196 --precision 3 > $O/step_times_p3.jsonl 2>> $O/bench.err
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $O/pmc_a -o pmc -- python3 bench.py --steps 3 --warmup 1 --precision 9 --no-probe --no-cpu-baseline --no-graph > $O/pmc_a.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_b -o pmc -- python3 bench.py --steps 3 --warmup 1 --precision 9 --no-probe --no-cpu-baseline --no-graph > $O/pmc_b.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_c -o pmc -- python3 bench.py --steps 3 --warmup 1 --precision 9 --no-probe --no-cpu-baseline --no-graph > $O/pmc_c.log 2>&1
# the same three passes for split-bf16 (precision 3: what "auto" runs on a checkpoint whose chain amplifies operand rounding)
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $O/pmc_a_p3 -o pmc -- python3 bench.py --steps 3 --warmup 1 --precision 3 --no-probe --no-cpu-baseline --no-graph > $O/pmc_a_p3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_b_p3 -o pmc -- python3 bench.py --steps 3 --warmup 1 --precision 3 --no-probe --no-cpu-baseline --no-graph > $O/pmc_b_p3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_c_p3 -o pmc -- python3 bench.py --steps 3 --warmup 1 --precision 3 --no-probe --no-cpu-baseline --no-graph > $O/pmc_c_p3.log 2>&1
# calibration of FETCH_SIZE / WRITE_SIZE on known-byte kernels of this library's access shapes (tools/microbench/fetch_calib.hip)
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/calib_f -o pmc -- ./tools/microbench/_bin/fetch_calib > $O/calib_f.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/calib_w -o pmc -- ./tools/microbench/_bin/fetch_calib > $O/calib_w.log 2>&1
# keep what travels back small: only the summaries
find $O -name "*_kernel_stats.csv" -o -name "*counter_collection.csv" | head -30 > $O/files.txt
find $O \( -name "*.db" -o -name "*_kernel_trace.csv" -o -name "*agent_info.csv" \) -delete
