# A/B of the split-bf16 long-window attention core (attention.h) against a variant build:  bash tools/experiments/p3_ab.sh <outdir> <tag>
mkdir -p gpurun_out/$1; O=gpurun_out/$1; TAG=$2
AB_PRECS=3 timeout 300 python tools/experiments/core_ab.py dump $O/new.pt > $O/ab.txt 2>&1
AB_PRECS=3 EGOEGO_PERFDEBUG_TAG=$TAG timeout 300 python tools/experiments/core_ab.py dump $O/old.pt >> $O/ab.txt 2>&1
python tools/experiments/core_ab.py cmp $O/new.pt $O/old.pt >> $O/ab.txt 2>&1; echo "cmp rc=$?" >> $O/ab.txt
rm -f $O/new.pt $O/old.pt
for rep in 1 2; do echo "== this build, T=120" >> $O/kt.txt; KT_PREC=3 timeout 200 python tools/kernel_times.py 2>&1 | grep -v amdgpu >> $O/kt.txt; echo "== variant $TAG, T=120" >> $O/kt.txt; KT_PREC=3 EGOEGO_PERFDEBUG_TAG=$TAG timeout 200 python tools/kernel_times.py 2>&1 | grep -v amdgpu >> $O/kt.txt; done
for rep in 1 2; do echo "== this build" >> $O/kt.txt; KT_PREC=3 KT_T=196 timeout 200 python tools/kernel_times.py 2>&1 | grep -v amdgpu >> $O/kt.txt; echo "== variant $TAG" >> $O/kt.txt; KT_PREC=3 KT_T=196 EGOEGO_PERFDEBUG_TAG=$TAG timeout 200 python tools/kernel_times.py 2>&1 | grep -v amdgpu >> $O/kt.txt; done
timeout 200 python tools/step_times.py --steps 100 --batches 1,32,256 --windows 120,196 --precision 3 2>&1 | grep -v amdgpu > $O/st_new.txt
EGOEGO_PERFDEBUG_TAG=$TAG timeout 200 python tools/step_times.py --steps 100 --batches 1,32,256 --windows 120,196 --precision 3 2>&1 | grep -v amdgpu > $O/st_old.txt
grep -c "equal True" $O/ab.txt; grep "equal False\|cmp rc\|rror" $O/ab.txt | head; grep "attn\|qkv\|sum\|==" $O/kt.txt; cat $O/st_new.txt $O/st_old.txt
