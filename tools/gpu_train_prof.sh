#!/bin/bash
# The training kernels alone (section 6 of tools/evidence.sh): tools/bench_backward.py, its kernel trace, and the HBM bytes /
# MfmaUtil of every training kernel from separate PMC passes.   gpurun --timeout 900 -- 'tools/gpu_train_prof.sh [tag]'
out=gpurun_out/${1:-train_prof}; mkdir -p $out
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - > /dev/null
python tools/bench_backward.py 2>&1 | grep -v amdgpu.ids > $out/bench_backward.txt
ONLY_NETS=1 ONLY_FUSED=1 timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $out/trace_backward -o p -- python tools/bench_backward.py > $out/trace_backward.log 2>&1
python tools/trace_top.py $out/trace_backward 16 > $out/trace_backward_top.txt 2>&1
for ctr in FETCH_SIZE WRITE_SIZE MfmaUtil; do
  ONLY_NETS=1 ONLY_FUSED=1 timeout 200 rocprofv3 --pmc $ctr -d $out/pmc_train/pmc_$ctr -o p -- python tools/bench_backward.py > /dev/null 2>&1
done
python tools/pmc_training.py $out/pmc_train > $out/pmc_training.md 2>&1
find $out -name "*.db" -delete; rm -rf $out/trace_backward $out/pmc_train
cat $out/bench_backward.txt $out/trace_backward_top.txt $out/pmc_training.md
