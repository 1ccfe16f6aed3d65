#!/bin/bash
# Round-2 measurement artifacts (run on the GPU box: gpurun -- 'bash tools/profile_round.sh').  Everything lands under
# gpurun_out/r02/; the summaries that are judged are copied into profiles/ by hand afterwards (tools/*.py below).
#   1. the bench line                                   python bench.py
#   2. per-kernel time summary                          rocprofv3 --kernel-trace --stats  (same command, fewer steps)
#   3. PMC counters, three separate passes              rocprofv3 --kernel-trace --pmc ...   (no --sys-trace etc.)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02; mkdir -p $O
python3 bench.py --steps 200 --warmup 10 > $O/bench_b256_t120.json 2> $O/bench.err
python3 bench.py --steps 200 --warmup 10 --batch 32 --no-cpu-baseline > $O/bench_b32_t120.json 2>> $O/bench.err
python3 bench.py --steps 100 --warmup 10 --batch 256 --window 196 --no-cpu-baseline > $O/bench_b256_t196.json 2>> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $O/pmc_a -o pmc -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph > $O/pmc_a.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_b -o pmc -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph > $O/pmc_b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_c -o pmc -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph > $O/pmc_c.log 2>&1
find $O -name "*.csv" | head -30 > $O/files.txt
