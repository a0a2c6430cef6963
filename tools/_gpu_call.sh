mkdir -p gpurun_out/r01q
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_f16x3.py -x -q -m gpu > gpurun_out/r01q/pytest.log 2>&1; echo rc=$? >> gpurun_out/r01q/pytest.log
ITERS=20 PRECISION=fp32 python tools/bench_mlp.py > gpurun_out/r01q/mlp_f32.log 2>&1
ITERS=40 PRECISION=fp16x3 python tools/bench_mlp.py > gpurun_out/r01q/mlp_h.log 2>&1
