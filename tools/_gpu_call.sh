mkdir -p gpurun_out/r01i
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r01i/pytest.log 2>&1; echo rc=$? >> gpurun_out/r01i/pytest.log
python bench.py --steps 1 --warmup 1 --cpu-baseline-rays 0 > gpurun_out/r01i/bench.log 2>&1; echo rc=$? >> gpurun_out/r01i/bench.log
