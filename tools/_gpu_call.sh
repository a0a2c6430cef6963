mkdir -p gpurun_out/r01j
python bench.py --steps 1 --warmup 1 --cpu-baseline-rays 0 --no-second-precision > gpurun_out/r01j/bench.log 2>&1; echo rc=$? >> gpurun_out/r01j/bench.log
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu > gpurun_out/r01j/pytest.log 2>&1; echo rc=$? >> gpurun_out/r01j/pytest.log
