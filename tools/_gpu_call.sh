mkdir -p gpurun_out/r01u
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r01u/pytest.log 2>&1; echo rc=$? >> gpurun_out/r01u/pytest.log
python bench.py --steps 1 --warmup 1 --cpu-baseline-rays 0 --no-second-precision > gpurun_out/r01u/bench.log 2>&1; echo rc=$? >> gpurun_out/r01u/bench.log
