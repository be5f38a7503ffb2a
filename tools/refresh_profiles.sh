#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/refresh_profiles.sh r01_f
# bench line with cpu_baseline, rocprofv3 kernel stats of the same command, and the two PMC passes for the roofline's `traffic`.
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python bench.py > gpurun_out/${tag}_bench.json
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -- python bench.py --no-cpu-baseline > gpurun_out/${tag}_prof.log 2>&1
echo "stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_fetch -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gemm-timing > gpurun_out/${tag}_fetch.log 2>&1
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_write -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gemm-timing > gpurun_out/${tag}_write.log 2>&1
echo "write done"
stats=$(ls gpurun_out/${tag}_prof/*/*kernel_stats.csv | head -1)
cp "$stats" gpurun_out/${tag}_bench_kernel_stats.csv
python tools/prof_summary.py gpurun_out/${tag}_bench_kernel_stats.csv gpurun_out/${tag}_bench_kernel_stats.md "${tag}: python bench.py --no-cpu-baseline (20 timed + 5 warm-up steps) under rocprofv3 --kernel-trace --stats" 25
python tools/pmc_traffic.py gpurun_out/${tag}_fetch gpurun_out/${tag}_write "gemm_nt4dma_kernel<true, true, 0, 0, false, false>" gpurun_out/${tag}_traffic.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gemm-timing"
tail -c 600 gpurun_out/${tag}_bench.json
