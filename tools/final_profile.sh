#!/bin/bash
# Round-end measurement pass on the GPU box (run through gpurun): parity suite, smoke, the default bench line, the
# rocprofv3 kernel trace of the same command and the two PMC passes.  Usage: tools/final_profile.sh <tag>
# Outputs land in gpurun_out/<tag>/; summarise with tools/rocpd_summary.py / tools/pmc_traffic.py into profiles/.
tag=${1:-final}
out=gpurun_out/$tag
mkdir -p $out
timeout 900 python -m pytest tests -x -q -m gpu > $out/pytest.log 2>&1; echo rc=$? >> $out/pytest.log
python __graft_entry__.py smoke > $out/smoke.log 2>&1; echo rc=$? >> $out/smoke.log
python bench.py --eager-gpu-baseline-rays 57344 > $out/bench.json 2> $out/bench.err; echo rc=$? >> $out/bench.err
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
CMD="python bench.py --steps 1 --warmup 0 --cpu-baseline-rays 0"
rocprofv3 --kernel-trace --stats -d $out/trace -o p -- $CMD > $out/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o p -- $CMD --no-second-precision > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/write -o p -- $CMD --no-second-precision > $out/write.log 2>&1
rocprofv3 --pmc MfmaUtil -d $out/mfma -o p -- $CMD > $out/mfma.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS -d $out/lds -o p -- $CMD --no-second-precision > $out/lds.log 2>&1
ls -R $out | head -40
