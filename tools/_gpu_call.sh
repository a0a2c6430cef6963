mkdir -p gpurun_out/r01l
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r01l/pytest.log 2>&1; echo rc=$? >> gpurun_out/r01l/pytest.log
