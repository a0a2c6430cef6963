#!/bin/bash
# shortest hardware A/B of the wave stage kernel: tree build vs libstnerf_hip_base.so, background items only, timing only
out=gpurun_out/${1:-wavet}
mkdir -p $out
export PYTHONDONTWRITEBYTECODE=1
for v in main base; do
  echo "== $v" >> $out/time.log
  if [ $v = base ]; then export STNERF_LIB=$PWD/st-nerf_amd/libstnerf_hip_base.so; else unset STNERF_LIB; fi
  KERNELS=wave CASES="bkgd" ITERS=3 timeout 40 python tools/ab_wave.py time 2>&1 | grep -v amdgpu.ids >> $out/time.log
done
cat $out/time.log
