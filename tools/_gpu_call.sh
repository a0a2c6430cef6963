set -x
mkdir -p gpurun_out/r01h
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 tools/check_rccl_single.py > gpurun_out/r01h/rccl.log 2>&1; echo rc=$? >> gpurun_out/r01h/rccl.log
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 1 --warmup 1 --cpu-baseline-rays 0 --no-second-precision > gpurun_out/r01h/bench_torchrun.log 2>&1; echo rc=$? >> gpurun_out/r01h/bench_torchrun.log
ITERS=80 PRECISION=fp32 tools/power_trace.sh gpurun_out/r01h/power_f32.csv python tools/bench_mlp.py > gpurun_out/r01h/mlp_f32.log 2>&1
ITERS=250 PRECISION=fp16x3 tools/power_trace.sh gpurun_out/r01h/power_f16x3.csv python tools/bench_mlp.py > gpurun_out/r01h/mlp_f16x3.log 2>&1
/opt/rocm/bin/rocm-smi --showpower --showclocks --csv > gpurun_out/r01h/idle.csv 2>&1
/opt/rocm/bin/rocm-smi --showmaxpower > gpurun_out/r01h/maxpower.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/mfma_rate tools/micro/mfma_rate.hip && tools/power_trace.sh gpurun_out/r01h/power_micro.csv /tmp/mfma_rate > gpurun_out/r01h/micro.log 2>&1
