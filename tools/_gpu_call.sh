mkdir -p gpurun_out/r01v
for w in taekwondo-1080p-90+30 single-512-64+64 walking-1080p-L4-64+64; do
  python bench.py --workload $w --steps 1 --warmup 1 --cpu-baseline-rays 0 > gpurun_out/r01v/$w.log 2>&1; echo rc=$? >> gpurun_out/r01v/$w.log
done
python bench.py --workload synthetic-4k-L8-128+64 --rays-per-launch 131072 --steps 1 --warmup 0 --cpu-baseline-rays 0 --no-second-precision > gpurun_out/r01v/c5.log 2>&1; echo rc=$? >> gpurun_out/r01v/c5.log
python bench.py --workload synthetic-4k-L8-128+64 --rays-per-launch 131072 --steps 1 --warmup 0 --cpu-baseline-rays 0 --no-second-precision --precision fp16x3 > gpurun_out/r01v/c5_h.log 2>&1; echo rc=$? >> gpurun_out/r01v/c5_h.log
