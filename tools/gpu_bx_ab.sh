#!/bin/bash
# A/B of the bf16x3 stage kernel on ONE box: the tree's build against variant builds (libstnerf_hip_<tag>.so), alternating
out=gpurun_out/${1:-bxab}; shift
mkdir -p $out
export PYTHONDONTWRITEBYTECODE=1
for round in 1 2 3; do
  for v in main "$@"; do
    if [ $v = main ]; then unset STNERF_LIB; else export STNERF_LIB=$PWD/st-nerf_amd/libstnerf_hip_$v.so; fi
    timeout 120 python tools/ab_bx.py 2>&1 | grep -v amdgpu.ids >> $out/time.log
  done
done
unset STNERF_LIB
cat $out/time.log
