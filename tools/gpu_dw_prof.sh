#!/bin/bash
# stnerf_train_dw_batch alone: timing, then HBM bytes / MfmaUtil per network from separate PMC passes.  gpurun -- 'tools/gpu_dw_prof.sh [tag]'
out=gpurun_out/${1:-dw_prof}; mkdir -p $out
export TMPDIR=/tmp
python tools/bench_dw.py 2>&1 | grep -v amdgpu.ids | tee $out/bench_dw.txt
for net in space motion; do
  for ctr in FETCH_SIZE WRITE_SIZE MfmaUtil; do
    ONLY=$net REPS=3 timeout 200 rocprofv3 --pmc $ctr -d $out/pmc_$net/pmc_$ctr -o p -- python tools/bench_dw.py > /dev/null 2>&1
  done
  echo "== $net" | tee -a $out/pmc.md; python tools/pmc_training.py $out/pmc_$net 2>&1 | grep -v "pack_\|encode" | tee -a $out/pmc.md
  ONLY=$net REPS=3 timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $out/trace_$net -o p -- python tools/bench_dw.py > /dev/null 2>&1
  python tools/trace_top.py $out/trace_$net 4 | tee -a $out/pmc.md
done
find $out -name "*.db" -delete; rm -rf $out/pmc_space $out/pmc_motion $out/trace_space $out/trace_motion
