#!/bin/bash
# Two data-parallel ranks sharing ONE GPU over gloo (a one-GPU box cannot run RCCL between two ranks): the bucketed exchange on the side stream, the
# deferred embedding bucket, the scalar communicator and the dynamic GEMM tile order with real HIP kernels; then bench.py's own --gpus 2 self-launch.
# usage (repo root, on the GPU box): bash tools/dp_rehearsal.sh <tag>
tag=${1:-dp2}
export SSI_LOCAL_DEVICE=0 SSI_DIST_BACKEND=gloo
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29611 \
    tests/workers/dp_step_worker.py --out gpurun_out/${tag}_worker.json > gpurun_out/${tag}_worker.log 2>&1
echo "worker rc=$?"; cat gpurun_out/${tag}_worker.json; echo
timeout -k 10 400 python bench.py --gpus 2 --steps 6 --warmup 2 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench rc=$?"; cut -c1-900 gpurun_out/${tag}_bench.json; tail -3 gpurun_out/${tag}_bench.err
