#!/bin/bash
# The round's evidence pass on the GPU box (run through gpurun; ~20 GPU-minutes): everything profiles/<rnd>_*.md is built from,
# measured on ONE build.  Usage: bash tools/evidence.sh [tag] [rnd] ; summaries land in gpurun_out/<tag>/summary/
tag=${1:-r06}
rnd=${2:-${tag:0:3}}
out=gpurun_out/$tag
mkdir -p $out
export PYTHONDONTWRITEBYTECODE=1
{ echo "build: $(ls -la --time-style=full-iso st-nerf_amd/libstnerf_hip.so)"; echo "rev: $(cat .git_rev 2>/dev/null)"; rocm-smi --showproductname 2>/dev/null | grep -i "card series\|gfx" | head -3; } > $out/env.txt
# ---- 1. parity suite + smoke
timeout 1500 python -m pytest tests -q -m gpu > $out/pytest.log 2>&1; echo rc=$? >> $out/pytest.log
timeout 200 python __graft_entry__.py smoke > $out/smoke.log 2>&1; echo rc=$? >> $out/smoke.log
# ---- 2. the driver's command with the driver's step counts, socket power / shader clock sampled throughout (>= 60 s per
#         arithmetic: 25 poses x ~3 s bf16x3, 25 x ~4.8 s f32); bf16x3 = the headline leg, fp32 = the co-equal second leg
timeout 900 bash tools/power_trace.sh $out/power_bench.csv python bench.py --steps 20 --warmup 5 --emulate-share 2,4,8 --detail-out $out/bench_detail.json > $out/bench.json 2> $out/bench.err; echo rc=$? >> $out/bench.err
# ---- 3. rocprofv3 of ONE step (pose 0) of the same workload per arithmetic: kernel trace, then one PMC pass per counter set
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
CMD="python bench.py --steps 1 --warmup 0 --cpu-baseline-rays 0 --eager-gpu-baseline-rays 0 --no-psnr-check --no-second-precision --no-config-legs --detail-out /tmp/d.json"
for prec in bf16x3 fp32; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $out/trace_$prec -o p -- $CMD --precision $prec > $out/trace_$prec.log 2>&1
  # (counters in their own runs, never together with a trace: gpurun refuses the combination)
  for ctr in FETCH_SIZE WRITE_SIZE MfmaUtil; do
    timeout 300 rocprofv3 --pmc $ctr -d $out/pmc_${ctr}_$prec -o p -- $CMD --precision $prec > $out/pmc_${ctr}_$prec.log 2>&1
  done
  if [ $prec = bf16x3 ]; then
    timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS -d $out/pmc_SQ_$prec -o p -- $CMD --precision $prec > $out/pmc_SQ_$prec.log 2>&1
  fi
done
# ---- 4. the other BASELINE configurations on the same build (one bench line each, both arithmetics)
W="--cpu-baseline-rays 0 --eager-gpu-baseline-rays 0 --no-psnr-check --no-config-legs"
timeout 300 python bench.py --workload single-512-64+64 --steps 5 --warmup 1 $W --detail-out $out/bench_c2_detail.json > $out/bench_c2.json 2> $out/bench_c2.err
timeout 300 python bench.py --workload taekwondo-1080p-90+30 --steps 3 --warmup 1 $W --detail-out $out/bench_c3_90_30_detail.json > $out/bench_c3_90_30.json 2> $out/bench_c3_90_30.err
timeout 500 python bench.py --workload walking-1080p-L4-64+64 --steps 2 --warmup 1 --emulate-share 2,4,8 $W --detail-out $out/bench_c4_detail.json > $out/bench_c4.json 2> $out/bench_c4.err
timeout 900 python bench.py --workload synthetic-4k-L8-128+64 --steps 1 --warmup 1 --rays-per-launch 131072 --emulate-share 2,4,8 $W --detail-out $out/bench_c5_detail.json > $out/bench_c5.json 2> $out/bench_c5.err
if [ -n "$EVIDENCE_FULL" ]; then
# ---- 4b. compositor / resampler at C4 and C5 on counter bytes: kernel trace + FETCH_SIZE + WRITE_SIZE of one step each
CMDX="python bench.py --steps 1 --warmup 0 --cpu-baseline-rays 0 --eager-gpu-baseline-rays 0 --no-psnr-check --no-second-precision --no-config-legs --precision bf16x3 --detail-out /tmp/d.json"
for cfg in "c4 walking-1080p-L4-64+64" "c5 synthetic-4k-L8-128+64 --rays-per-launch 131072"; do
  set -- $cfg; tagc=$1; shift
  timeout 400 rocprofv3 --kernel-trace --stats -d $out/trace_$tagc -o p -- $CMDX --workload "$@" > $out/trace_$tagc.log 2>&1
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --pmc $ctr -d $out/pmc_${ctr}_$tagc -o p -- $CMDX --workload "$@" > $out/pmc_${ctr}_$tagc.log 2>&1
  done
done
fi
# ---- 5. N ranks on this one GPU through bench.py itself (gloo; the code path a node runs, not a measurement)
timeout 300 python bench.py --gpus 2 --debug-single-device --steps 2 --warmup 1 $W --no-second-precision --detail-out $out/bench_2ranks_one_device_detail.json > $out/bench_2ranks_one_device.json 2> $out/bench_2ranks_one_device.err
# ---- 6. the training kernels: GEMM flavours, whole backward, kernel trace
timeout 200 python tools/bench_backward.py > $out/bench_backward.txt 2>&1
ONLY_NETS=1 ONLY_FUSED=1 timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $out/trace_backward -o p -- python tools/bench_backward.py > $out/trace_backward.log 2>&1
python tools/trace_top.py $out/trace_backward 30 > $out/trace_backward_top.txt 2>&1
for ctr in FETCH_SIZE WRITE_SIZE MfmaUtil; do
  ONLY_NETS=1 ONLY_FUSED=1 timeout 200 rocprofv3 --pmc $ctr -d $out/pmc_train/pmc_$ctr -o p -- python tools/bench_backward.py > /dev/null 2>&1
done
python tools/pmc_training.py $out/pmc_train > $out/pmc_training.md 2>&1
# stnerf_train_dw_batch alone, on operands laid out as the autograd hands them over: timing, then HBM bytes / MfmaUtil per network
bash tools/gpu_dw_prof.sh $tag/dw > $out/dw_prof.log 2>&1
# one iteration of the reference trainer's inner loop: the bench scene at 4096 rays, the reference's own batch configuration, both against eager PyTorch-ROCm
{ timeout 300 python tools/bench_train_step.py --iters 12 --eager; timeout 300 python tools/bench_train_step.py --iters 12 --rays 2000 --workload taekwondo-1080p-90+30 --eager;
  timeout 300 python tools/bench_train_step.py --iters 8 --rays 16384; } 2>&1 | grep -v "Warning\|warn\|amdgpu.ids\|float(loss)" > $out/train_step.txt
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $out/trace_step -o p -- python tools/bench_train_step.py --iters 3 > $out/trace_step.log 2>&1
python tools/trace_top.py $out/trace_step 30 > $out/trace_step_top.txt 2>&1
# ---- 7. stage kernels alone, A/B of the bf16x3 variants, compositor and resampler alone, microbenchmarks
STAGE_ONLY=1 timeout 200 python tools/bench_stage.py > $out/bench_stage.txt 2>&1
timeout 200 python tools/bench_composite.py > $out/bench_composite.txt 2>&1
timeout 200 python tools/resample_phase_prof.py > $out/bench_resample.txt 2>&1
timeout 100 tools/micro/hbm_copy > $out/hbm_copy.json 2>/dev/null
# ---- 8. summarise here (the rocprofv3 databases are too large to travel), then drop them
python tools/summarise.py $out $rnd > $out/summarise.log 2>&1; echo rc=$? >> $out/summarise.log
find $out -name "*.db" -delete; find $out -type d -empty -delete
rm -rf $out/trace_*/ $out/pmc_*/ 2>/dev/null
tail -3 $out/pytest.log; tail -2 $out/smoke.log; tail -40 $out/summarise.log; for f in c2 c3_90_30 c4 c5 2ranks_one_device; do tail -c 200 $out/bench_$f.err; done
du -sh $out
