mkdir -p gpurun_out/r01o
tools/power_trace.sh gpurun_out/r01o/power_micro.csv tools/micro/mfma_rate sustained > gpurun_out/r01o/micro.log 2>&1
