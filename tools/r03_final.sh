#!/bin/bash
# usage (GPU box, repo root): bash tools/r03_final.sh <tag>  — the judged artefacts of the round (bench line with cpu_baseline, rocprofv3 kernel
# stats of the same command, FETCH / WRITE passes -> roofline.traffic) plus the secondary workloads of BASELINE.json at one GPU.
tag=$1
bash tools/refresh_profiles.sh $tag || exit 1
cd "$GRAFT_REPO_ROOT"
python bench.py --no-cpu-baseline --n-dsus 8192 > gpurun_out/${tag}_dsus8192_bench.json 2>/dev/null; echo "A' rc=$?"
python bench.py --no-cpu-baseline --seq 4096 > gpurun_out/${tag}_s4096_bench.json 2>/dev/null; echo "C rc=$?"
python bench.py --no-cpu-baseline --packed --seq 8192 --batch 2 --n-dsus 2048 > gpurun_out/${tag}_packed_bench.json 2>/dev/null; echo "E rc=$?"
python bench.py --no-cpu-baseline --padded > gpurun_out/${tag}_padded_bench.json 2>/dev/null; echo "padded rc=$?"
for f in dsus8192 s4096 packed padded; do python - <<PY
import json
d=json.load(open("gpurun_out/${tag}_${f}_bench.json"))
print("${f}", round(d["value"]), "tokens/s", round(d["ms_per_step"],2), "ms", d.get("mfma_roofline_frac_step"))
PY
done
