export PYTHONDONTWRITEBYTECODE=1
out=gpurun_out/cm9; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_composite_merge.py tests/test_gpu_ops.py -x -q -k "composite or merge" > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log
timeout 300 python tools/bench_composite.py > $out/bench_merge.log 2>&1
STNERF_LIB=$PWD/st-nerf_amd/libstnerf_hip_cprof.so timeout 200 python tools/comp_phase_prof.py > $out/phases_l3.log 2>&1
STNERF_LIB=$PWD/st-nerf_amd/libstnerf_hip_cprof.so L=9 S=192,128 timeout 200 python tools/comp_phase_prof.py > $out/phases_l9.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
CMD="python bench.py --steps 1 --warmup 0 --cpu-baseline-rays 0 --eager-gpu-baseline-rays 0 --no-psnr-check --no-second-precision --precision bf16x3"
for w in taekwondo-1080p-64+64 walking-1080p-L4-64+64 synthetic-4k-L8-128+64; do
  extra=""; [ $w = synthetic-4k-L8-128+64 ] && extra="--rays-per-launch 131072"
  timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $out/tr_$w -o p -- $CMD --workload $w $extra > $out/tr_$w.log 2>&1
  echo "== $w" >> $out/stats.txt
  find $out/tr_$w -name "*kernel_stats.csv" | head -1 | xargs cat | grep -i "composite\|resample_k\|sample_coarse" | cut -c1-110 >> $out/stats.txt
done
CASES="C3 fine 3x128" timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -f csv -d $out/pmc1 -o p -- python tools/bench_composite.py > $out/pmc1.log 2>&1
python tools/_pmc_sum.py $out/pmc1 > $out/pmc1.txt 2>&1
find $out -name "*kernel_trace.csv" -delete; find $out -name "*.db" -delete; find $out -name "*counter_collection.csv" -delete
tail -5 $out/tests.log; cat $out/bench_merge.log $out/phases_l3.log $out/phases_l9.log $out/stats.txt $out/pmc1.txt | grep -v amdgpu.ids
