#!/bin/bash
# Late addendum to tools/r04_evidence.sh (run through gpurun, ~9 GPU-minutes): after the evidence pass the compositor's insertion
# search changed (render.hip) and the per-network MLP kernels lost their A/B tile variants (mlp.hip); the stage, sampler,
# resampler and training kernels are the evidence pass's, instruction for instruction (profiles/r04_addendum_compositor.md).
# So: the whole parity suite + smoke on the final build, a short driver line (both arithmetics, the HBM-side kernels' times),
# one traced + counted step of C3, one traced step of C5, the compositor alone.
out=gpurun_out/${1:-r04j}
mkdir -p $out
export PYTHONDONTWRITEBYTECODE=1
{ echo "build: $(ls -la --time-style=full-iso st-nerf_amd/libstnerf_hip.so)"; echo "rev: $(cat .git_rev 2>/dev/null)"; } > $out/env.txt
timeout 1200 python -m pytest tests -q -m gpu > $out/pytest.log 2>&1; echo rc=$? >> $out/pytest.log
timeout 200 python __graft_entry__.py smoke > $out/smoke.log 2>&1; echo rc=$? >> $out/smoke.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-config-legs > $out/bench.json 2> $out/bench.err; echo rc=$? >> $out/bench.err
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
CMD="python bench.py --steps 1 --warmup 0 --cpu-baseline-rays 0 --eager-gpu-baseline-rays 0 --no-psnr-check --no-second-precision --no-config-legs --precision bf16x3"
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $out/trace -o p -- $CMD > $out/trace.log 2>&1
for ctr in SQ_INSTS_VALU FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $ctr -f csv -d $out/pmc_$ctr -o p -- $CMD > $out/pmc_$ctr.log 2>&1
done
timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $out/trace_c5 -o p -- $CMD --workload synthetic-4k-L8-128+64 --rays-per-launch 131072 > $out/trace_c5.log 2>&1
timeout 200 python tools/bench_composite.py > $out/bench_composite.txt 2>&1
python tools/r04_addendum.py $out > $out/summary.md 2>&1
find $out -name "*counter_collection.csv" -delete; find $out -name "*kernel_trace.csv" -delete; find $out -name "*.db" -delete
cat $out/summary.md
