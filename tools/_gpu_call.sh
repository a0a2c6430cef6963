mkdir -p gpurun_out/r01m
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r01m/pytest.log 2>&1; echo rc=$? >> gpurun_out/r01m/pytest.log
