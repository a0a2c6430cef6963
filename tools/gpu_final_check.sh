#!/bin/bash
# end-of-round validation on the GPU box: full GPU suite, smoke, headline bench, kernel trace of one frame
out=gpurun_out/${1:-final}
mkdir -p $out
export PYTHONDONTWRITEBYTECODE=1
timeout 400 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "rc=$?" >> $out/pytest.log
timeout 120 python __graft_entry__.py smoke > $out/smoke.log 2>&1; echo "rc=$?" >> $out/smoke.log
timeout 300 python bench.py --steps 5 --warmup 2 > $out/bench.json 2> $out/bench.err; echo "rc=$?" >> $out/bench.err
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 200 rocprofv3 --kernel-trace --stats -d $out/trace -o p -- python bench.py --steps 1 --warmup 0 --cpu-baseline-rays 0 --no-psnr-check > $out/trace.log 2>&1
tail -3 $out/pytest.log; tail -4 $out/smoke.log; cat $out/bench.json; tail -2 $out/bench.err
