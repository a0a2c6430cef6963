mkdir -p gpurun_out/r01k
timeout 900 python -m pytest tests/test_gpu_render.py -x -q -m gpu > gpurun_out/r01k/pytest.log 2>&1; echo rc=$? >> gpurun_out/r01k/pytest.log
timeout 600 python bench.py --steps 1 --warmup 1 --cpu-baseline-rays 0 --no-second-precision --eager-gpu-baseline-rays 14336 > gpurun_out/r01k/bench.log 2>&1; echo rc=$? >> gpurun_out/r01k/bench.log
