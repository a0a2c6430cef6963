#!/bin/bash
# Round-3 evidence pass on the GPU box (run through gpurun; ~20 GPU-minutes): everything profiles/r03_*.md is built from,
# measured on ONE build.  Usage: bash tools/r03_evidence.sh [tag] ; then python tools/r03_summarise.py gpurun_out/<tag>
tag=${1:-r03}
out=gpurun_out/$tag
mkdir -p $out
export PYTHONDONTWRITEBYTECODE=1
git_rev=$(cat .git_rev 2>/dev/null)
{ echo "build: $(ls -la --time-style=full-iso st-nerf_amd/libstnerf_hip.so)"; echo "rev: $git_rev"; rocm-smi --showproductname 2>/dev/null | grep -i "card series\|gfx" | head -3; } > $out/env.txt
# ---- 1. parity suite + smoke
timeout 900 python -m pytest tests -x -q -m gpu > $out/pytest.log 2>&1; echo rc=$? >> $out/pytest.log
timeout 200 python __graft_entry__.py smoke > $out/smoke.log 2>&1; echo rc=$? >> $out/smoke.log
# ---- 2. the driver's command (C3 64+64: headline f32, other_precision = bf16x3, other_precision_2 = fp16x3, CPU + eager legs)
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err; echo rc=$? >> $out/bench.err
# ---- 3. rocprofv3 of ONE step (pose 0) of the same workload, f32 then bf16x3: kernel trace, then one PMC pass per counter set
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
CMD="python bench.py --steps 1 --warmup 0 --cpu-baseline-rays 0 --eager-gpu-baseline-rays 0 --no-psnr-check --no-second-precision"
for prec in fp32 bf16x3; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $out/trace_$prec -o p -- $CMD --precision $prec > $out/trace_$prec.log 2>&1
  for ctr in FETCH_SIZE WRITE_SIZE MfmaUtil; do
    timeout 300 rocprofv3 --pmc $ctr -d $out/pmc_${ctr}_$prec -o p -- $CMD --precision $prec > $out/pmc_${ctr}_$prec.log 2>&1
  done
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS -d $out/pmc_SQ_$prec -o p -- $CMD --precision $prec > $out/pmc_SQ_$prec.log 2>&1
done
# ---- 4. the other BASELINE configurations on the same build (bench line each; f32 headline + bf16x3 leg)
W="--cpu-baseline-rays 0 --eager-gpu-baseline-rays 0 --no-psnr-check"
timeout 300 python bench.py --workload single-512-64+64 --steps 5 --warmup 1 $W > $out/bench_c2.json 2> $out/bench_c2.err
timeout 300 python bench.py --workload taekwondo-1080p-90+30 --steps 3 --warmup 1 $W > $out/bench_c3_90_30.json 2> $out/bench_c3_90_30.err
timeout 400 python bench.py --workload walking-1080p-L4-64+64 --steps 2 --warmup 1 $W > $out/bench_c4.json 2> $out/bench_c4.err
timeout 900 python bench.py --workload synthetic-4k-L8-128+64 --steps 1 --warmup 1 --rays-per-launch 131072 $W > $out/bench_c5.json 2> $out/bench_c5.err
# ---- 4b. compositor / resampler at C4 and C5 against the measured HBM rates: kernel trace + FETCH_SIZE + WRITE_SIZE of one step each
CMDX="python bench.py --steps 1 --warmup 0 --cpu-baseline-rays 0 --eager-gpu-baseline-rays 0 --no-psnr-check --no-second-precision --precision bf16x3"
for cfg in "c4 walking-1080p-L4-64+64" "c5 synthetic-4k-L8-128+64 --rays-per-launch 131072"; do
  set -- $cfg; tagc=$1; shift
  timeout 400 rocprofv3 --kernel-trace --stats -d $out/trace_$tagc -o p -- $CMDX --workload "$@" > $out/trace_$tagc.log 2>&1
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --pmc $ctr -d $out/pmc_${ctr}_$tagc -o p -- $CMDX --workload "$@" > $out/pmc_${ctr}_$tagc.log 2>&1
  done
done
# ---- 5. stage kernels alone (with socket power / clock sampled by rocm-smi), compositor and resampler alone, microbenchmarks
STAGE_ONLY=1 CASES="bf16x3" ITERS=200 timeout 200 bash tools/power_trace.sh $out/power_bf16x3.csv python tools/bench_stage.py > $out/power_bf16x3.log 2>&1
STAGE_ONLY=1 CASES="stage" ITERS=40 timeout 200 bash tools/power_trace.sh $out/power_f32.csv python tools/bench_stage.py > $out/power_f32.log 2>&1
timeout 200 python tools/bench_composite.py > $out/bench_composite.txt 2>&1
STNERF_COMPOSITE_KERNEL=staged timeout 200 python tools/bench_composite.py > $out/bench_composite_staged.txt 2>&1
timeout 200 python tools/resample_phase_prof.py > $out/bench_resample.txt 2>&1
STAGE_ONLY=1 timeout 200 python tools/bench_stage.py > $out/bench_stage.txt 2>&1
timeout 100 tools/micro/bf16x3_proto > $out/bf16x3_proto.txt 2>&1
timeout 100 tools/micro/hbm_copy > $out/hbm_copy.json 2>/dev/null
# ---- 6. summarise here (the rocprofv3 databases are too large to travel: gpurun merges at most 64 MiB back), then drop them
python tools/r03_summarise.py $out > $out/summarise.log 2>&1; echo rc=$? >> $out/summarise.log
find $out -name "*.db" -delete; find $out -type d -empty -delete
rm -rf $out/trace_fp32 $out/trace_bf16x3 $out/trace_c4 $out/trace_c5 $out/pmc_*_fp32 $out/pmc_*_bf16x3 $out/pmc_*_c4 $out/pmc_*_c5 2>/dev/null
tail -3 $out/pytest.log; tail -3 $out/smoke.log; tail -30 $out/summarise.log; for f in c2 c3_90_30 c4 c5; do tail -c 200 $out/bench_$f.err; done
du -sh $out
