#!/bin/bash
# per-kernel time summaries only (rocprofv3 --kernel-trace --stats) at B=256 and B=32: a quick look between full profile rounds
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/quick; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_b32 -o stats -- python3 bench.py --steps 50 --warmup 3 --batch 32 --no-cpu-baseline > $O/stats_b32.log 2>&1
for d in stats stats_b32; do echo $d; python3 - $O/$d <<'PY'
import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:5]:
    print('  %-100s calls %5s avg %8.1f us  %5s%%'%(r['Name'][:100], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
done
find $O \( -name "*.db" -o -name "*_kernel_trace.csv" -o -name "*agent_info.csv" \) -delete
