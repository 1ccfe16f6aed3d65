# rocprofv3 kernel statistics of the step at the three measured shapes (run on the GPU box through gpurun)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for cfg in "t120 --batches 256 --steps 50" "t196 --batches 256 --windows 196 --steps 30" "b32 --batches 32 --steps 100"; do
  set -- $cfg; tag=$1; shift
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o $tag -- python3 tools/step_times.py "$@" > gpurun_out/prof_$tag.log 2>&1
done
