#!/bin/bash
# First hardware trial of csrc/mlp_wave.hip (run through gpurun): bitwise A/B against the LDS stage kernel, per-layer
# localisation with the development build, timing; the GPU test suite + a short bench on the wave kernel only if the A/B holds.
out=gpurun_out/${1:-wave1}
mkdir -p $out
export PYTHONDONTWRITEBYTECODE=1
timeout 300 python tools/ab_wave.py check > $out/check.log 2>&1; echo "rc=$?" >> $out/check.log
STNERF_LIB=$PWD/st-nerf_amd/libstnerf_hip_dbg.so timeout 300 python tools/ab_wave.py layers > $out/layers.log 2>&1; echo "rc=$?" >> $out/layers.log
timeout 240 python tools/ab_wave.py time > $out/time.log 2>&1; echo "rc=$?" >> $out/time.log
if grep -q "CHECK OK" $out/check.log; then
  STNERF_STAGE_KERNEL=wave timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest_wave.log 2>&1; echo "rc=$?" >> $out/pytest_wave.log
  STNERF_STAGE_KERNEL=wave timeout 300 python bench.py --steps 2 --warmup 1 --cpu-baseline-rays 0 --no-second-precision --no-psnr-check > $out/bench_wave.json 2> $out/bench_wave.err
  STNERF_STAGE_KERNEL=lds timeout 300 python bench.py --steps 2 --warmup 1 --cpu-baseline-rays 0 --no-second-precision --no-psnr-check > $out/bench_lds.json 2> $out/bench_lds.err
fi
tail -n 40 $out/check.log $out/layers.log $out/time.log
tail -n 5 $out/pytest_wave.log 2>/dev/null
head -c 600 $out/bench_wave.json 2>/dev/null; echo; head -c 600 $out/bench_lds.json 2>/dev/null
