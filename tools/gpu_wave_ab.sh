#!/bin/bash
# hardware A/B of the wave stage kernel against a variant build (libstnerf_hip_base.so = the previous source, built with
# STNERF_LIB_TAG=base): bitwise check of the tree's build, alternating timings on the same box, phase profile
out=gpurun_out/${1:-waveab}
mkdir -p $out
export PYTHONDONTWRITEBYTECODE=1
STNERF_LIB=$PWD/st-nerf_amd/libstnerf_hip_base.so timeout 200 python tools/ab_wave.py check --save /tmp/base.pt > $out/check_base.log 2>&1
timeout 200 python tools/ab_wave.py check --against /tmp/base.pt > $out/check.log 2>&1; echo "rc=$?" >> $out/check.log
for v in main base main base; do
  echo "== $v" >> $out/time.log
  if [ $v = base ]; then export STNERF_LIB=$PWD/st-nerf_amd/libstnerf_hip_base.so; else unset STNERF_LIB; fi
  KERNELS=wave CASES="bkgd,performer fused" ITERS=3 timeout 100 python tools/ab_wave.py time >> $out/time.log 2>&1
done
if [ -f st-nerf_amd/libstnerf_hip_prof.so ]; then
  STNERF_LIB=$PWD/st-nerf_amd/libstnerf_hip_prof.so timeout 100 python tools/wave_prof.py > $out/prof.log 2>&1
fi
cat $out/check.log $out/time.log $out/prof.log | grep -v amdgpu.ids
