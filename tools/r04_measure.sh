#!/bin/bash
# usage (GPU box, repo root): bash tools/r04_measure.sh <tag>
# The judged artefacts of round 4: headline bench line with cpu_baseline, rocprofv3 kernel stats of the same command, FETCH / WRITE passes ->
# roofline.traffic (tools/refresh_profiles.sh); the secondary BASELINE.json workloads at one GPU; the reference's default micro-batches
# (conf/data/_sft_base.yaml:21: 2 x 2048; _cpt_base.yaml:23: 16 x 768); right-padded batches with and without the host-side unpadding; and
# the per-kernel counter table (matrix-pipe busy, traffic, clocks) from three separate --pmc passes.
tag=$1
bash tools/refresh_profiles.sh $tag || exit 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { name=$1; shift; python bench.py --no-cpu-baseline "$@" > gpurun_out/${tag}_${name}_bench.json 2>/dev/null; echo "$name rc=$?"; }
run dsus8192 --n-dsus 8192
run s4096 --seq 4096
run packed --packed --seq 8192 --batch 2 --n-dsus 2048
run padded --padded
run padded_as_is --padded --no-unpad
run b2_s2048 --batch 2 --seq 2048
run b16_s768 --batch 16 --seq 768
run sft_default_b2_s2048_ga4 --batch 2 --seq 2048 --grad-accum 4 --steps 10
run cpt_default_b16_s768_ga4 --batch 16 --seq 768 --grad-accum 4 --steps 6 --warmup 2
for f in dsus8192 s4096 packed padded padded_as_is b2_s2048 b16_s768 sft_default_b2_s2048_ga4 cpt_default_b16_s768_ga4; do python - <<PY
import json
d=json.load(open("gpurun_out/${tag}_${f}_bench.json"))
print("${f}", round(d["value"]), "tokens/s", round(d["ms_per_step"],2), "ms", d.get("mfma_roofline_frac_step"))
PY
done
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gemm-timing"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${tag}_sq -- $CMD > gpurun_out/${tag}_sq.log 2>&1; echo "sq rc=$?"
python tools/pmc_table.py gpurun_out/${tag}_fetch gpurun_out/${tag}_write gpurun_out/${tag}_sq gpurun_out/${tag}_pmc_kernels.md "${tag}: per-kernel counters of \`$CMD\` at HEAD" 16 > /dev/null; echo "table rc=$?"
python tools/pmc_table.py gpurun_out/${tag}_fetch gpurun_out/${tag}_write gpurun_out/${tag}_sq gpurun_out/${tag}_attn_pmc.md "${tag}: attention kernels, counters of \`$CMD\` at HEAD" 4 attn_ > /dev/null
cat gpurun_out/${tag}_attn_pmc.md | cut -c1-260
rm -rf gpurun_out/${tag}_fetch/*/*agent_info.csv
