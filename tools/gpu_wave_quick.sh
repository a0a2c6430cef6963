#!/bin/bash
# quick hardware check of the wave stage kernel: bitwise A/B, timing of both kernels, phase profile (development build)
out=gpurun_out/${1:-waveq}
mkdir -p $out
export PYTHONDONTWRITEBYTECODE=1
timeout 300 python tools/ab_wave.py check > $out/check.log 2>&1; echo "rc=$?" >> $out/check.log
ITERS=3 timeout 240 python tools/ab_wave.py time > $out/time.log 2>&1; echo "rc=$?" >> $out/time.log
if [ -f st-nerf_amd/libstnerf_hip_prof.so ]; then
  STNERF_LIB=$PWD/st-nerf_amd/libstnerf_hip_prof.so timeout 120 python tools/wave_prof.py > $out/prof.log 2>&1
fi
for tag in $VARIANTS; do
  echo "== variant '${tag}'" >> $out/time.log
  KERNELS=wave CASES="bkgd,performer fused" ITERS=3 STNERF_LIB=$PWD/st-nerf_amd/libstnerf_hip_${tag}.so timeout 120 python tools/ab_wave.py time >> $out/time.log 2>&1
done
cat $out/check.log $out/time.log $out/prof.log
