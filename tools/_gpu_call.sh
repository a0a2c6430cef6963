mkdir -p gpurun_out/r01t
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_render.py -x -q -m gpu > gpurun_out/r01t/pytest.log 2>&1; echo rc=$? >> gpurun_out/r01t/pytest.log
python bench.py --steps 1 --warmup 1 --cpu-baseline-rays 0 > gpurun_out/r01t/bench.log 2>&1; echo rc=$? >> gpurun_out/r01t/bench.log
