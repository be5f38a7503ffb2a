#!/bin/bash
# usage (GPU box, repo root): bash tools/r05_prof.sh <tag> <bench.py flags...>: rocprofv3 kernel stats of one bench.py workload -> gpurun_out/<tag>_kernel_stats.{csv,md}
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -- python bench.py --no-cpu-baseline --no-gemm-timing --steps 6 --warmup 2 "$@" > gpurun_out/${tag}_prof.log 2>&1 || { tail -20 gpurun_out/${tag}_prof.log; exit 1; }
stats=$(ls gpurun_out/${tag}_prof/*/*kernel_stats.csv | head -1)
cp "$stats" gpurun_out/${tag}_kernel_stats.csv
python tools/prof_summary.py gpurun_out/${tag}_kernel_stats.csv gpurun_out/${tag}_kernel_stats.md "${tag}: python bench.py --no-cpu-baseline --no-gemm-timing --steps 6 --warmup 2 $* under rocprofv3 --kernel-trace --stats" 8
rm -rf gpurun_out/${tag}_prof
grep -i "attn\|nt4_splitk_reduce\|Total" gpurun_out/${tag}_kernel_stats.md | cut -c1-200
