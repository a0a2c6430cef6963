mkdir -p gpurun_out/r01p
tools/power_trace.sh gpurun_out/r01p/power_dual.csv tools/micro/dual_issue > gpurun_out/r01p/dual.log 2>&1
