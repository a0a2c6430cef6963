mkdir -p gpurun_out/r01s
timeout 600 python -m pytest tests/test_gpu_render.py -x -q -m gpu -k "sharded or ragged or chunking" > gpurun_out/r01s/pytest.log 2>&1; echo rc=$? >> gpurun_out/r01s/pytest.log
STNERF_TILE_H=64 timeout 300 python -m pytest tests/test_gpu_f16x3.py -x -q -m gpu -k "vs_fp64" > gpurun_out/r01s/tileh64.log 2>&1; echo rc=$? >> gpurun_out/r01s/tileh64.log
python bench.py --steps 1 --warmup 1 --cpu-baseline-rays 0 --no-second-precision --partition stripes > gpurun_out/r01s/bench_stripes.log 2>&1; echo rc=$? >> gpurun_out/r01s/bench_stripes.log
