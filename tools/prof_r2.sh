cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_t120 -o t120 -- python3 tools/step_times.py --batches 256 --steps 50 > gpurun_out/prof_t120.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_t196 -o t196 -- python3 tools/step_times.py --batches 256 --windows 196 --steps 30 > gpurun_out/prof_t196.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_b32 -o b32 -- python3 tools/step_times.py --batches 32 --steps 100 > gpurun_out/prof_b32.log 2>&1
