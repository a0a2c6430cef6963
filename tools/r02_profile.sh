#!/bin/bash
# Round-2 measurement pass on the GPU box (run through gpurun).  Usage: bash tools/r02_profile.sh <tag>
# Outputs land in gpurun_out/<tag>/; summarise with tools/rocpd_summary.py / tools/pmc_traffic.py into profiles/.
tag=${1:-r02prof}
out=gpurun_out/$tag
mkdir -p $out
export PYTHONDONTWRITEBYTECODE=1
python bench.py --steps 5 --warmup 2 > $out/bench.json 2> $out/bench.err; echo rc=$? >> $out/bench.err
STNERF_STAGE_KERNEL=lds python bench.py --steps 3 --warmup 1 --cpu-baseline-rays 0 --no-second-precision --no-psnr-check > $out/bench_lds.json 2> $out/bench_lds.err
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
CMD="python bench.py --steps 1 --warmup 0 --cpu-baseline-rays 0 --no-psnr-check"
rocprofv3 --kernel-trace --stats -d $out/trace -o p -- $CMD > $out/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o p -- $CMD --no-second-precision > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/write -o p -- $CMD --no-second-precision > $out/write.log 2>&1
rocprofv3 --pmc MfmaUtil -d $out/mfma -o p -- $CMD --no-second-precision > $out/mfma.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA -d $out/sq -o p -- $CMD --no-second-precision > $out/sq.log 2>&1
ls -R $out | head -40
