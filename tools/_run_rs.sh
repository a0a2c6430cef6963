export PYTHONDONTWRITEBYTECODE=1
out=gpurun_out/rs4; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "resample" > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log
timeout 300 python tools/resample_phase_prof.py > $out/bench.log 2>&1
tail -4 $out/tests.log; cat $out/bench.log | grep -v amdgpu.ids
