#!/bin/bash
# usage (GPU box, repo root): bash tools/r05_measure.sh <tag> [headline | headline-only]   (headline-only: without the secondary lines)
# Round 5: the secondary workloads of one box in ONE file, gpurun_out/<tag>_secondary.json — every line with the attention backward kernels it
# ran (bench.py's "attention_backward": ssi_attn_last_dispatch) so that a reader sees which path a number came from.  With "headline" also the
# judged artefacts of the headline (tools/refresh_profiles.sh: bench line with cpu_baseline, rocprofv3 kernel stats, FETCH / WRITE passes).
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
HL=0; if [ "$2" = "headline" ] || [ "$2" = "headline-only" ]; then HL=1; fi
if [ $HL = 1 ]; then bash tools/refresh_profiles.sh $tag || exit 1; fi
if [ "$2" != "headline-only" ]; then
run() { name=$1; shift; python bench.py --no-cpu-baseline "$@" > gpurun_out/${tag}_sec_${name}.json 2>/dev/null; echo "$name rc=$?"; }
run headline
run dsus8192 --n-dsus 8192
run s4096_b8 --seq 4096
run packed_s8192_b2 --packed --seq 8192 --batch 2 --n-dsus 2048
run packed_s8192_b2_no_plan --packed --seq 8192 --batch 2 --n-dsus 2048 --no-attn-plan
run padded --padded
run padded_no_plan --padded --no-attn-plan
run padded_as_is --padded --no-unpad
run sft_default_b2_s2048 --batch 2 --seq 2048
run cpt_default_b16_s768 --batch 16 --seq 768
run sft_default_b2_s2048_ga4 --batch 2 --seq 2048 --grad-accum 4 --steps 10
run cpt_default_b16_s768_ga4 --batch 16 --seq 768 --grad-accum 4 --steps 6 --warmup 2
python - <<PY
import json, glob, os
out = {"what": "python bench.py --no-cpu-baseline <flags> on ONE box, back to back (tools/r05_measure.sh); 'attention_backward' = the kernels of the last micro-batch (ssi_attn_last_dispatch)", "lines": {}}
for f in sorted(glob.glob("gpurun_out/${tag}_sec_*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        out["lines"][os.path.basename(f)] = {"error": repr(e)}
        continue
    name = os.path.basename(f)[len("${tag}_sec_"):-5]
    out["lines"][name] = {"tokens_per_s": round(d["value"]), "ms_per_step": round(d["ms_per_step"], 2), "mfma_roofline_frac_step": round(d.get("mfma_roofline_frac_step", 0), 4),
                          "gflop_per_token": d.get("gflop_per_token"), "workload": d["config"]["workload"], "attention_backward": d.get("attention_backward")}
    print(f"{name:32s} {d['value']:9.0f} tok/s {d['ms_per_step']:8.2f} ms  {d.get('mfma_roofline_frac_step', 0):.4f}  {d.get('attention_backward')}")
    os.remove(f)
json.dump(out, open("gpurun_out/${tag}_secondary.json", "w"), indent=1)
PY
fi
if [ $HL = 1 ]; then
# per-kernel counters (three separate --pmc passes, never combined with trace domains other than the kernel trace): the headline's table, and the
# attention kernels of the right-padded workload (document-aware forms of the pipelined backward)
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gemm-timing"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${tag}_sq -- $CMD > gpurun_out/${tag}_sq.log 2>&1; echo "sq rc=$?"
python tools/pmc_table.py gpurun_out/${tag}_fetch gpurun_out/${tag}_write gpurun_out/${tag}_sq gpurun_out/${tag}_pmc_kernels.md "${tag}: per-kernel counters of \`$CMD\` at HEAD" 16 > /dev/null; echo "table rc=$?"
for pass in fetch:FETCH_SIZE write:WRITE_SIZE sq:"SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d gpurun_out/${tag}_pad_$name -- $CMD --padded > gpurun_out/${tag}_pad_$name.log 2>&1; echo "padded $name rc=$?"
done
python tools/pmc_table.py gpurun_out/${tag}_pad_fetch gpurun_out/${tag}_pad_write gpurun_out/${tag}_pad_sq gpurun_out/${tag}_attn_pmc.md "${tag}: attention kernels of \`$CMD --padded\` (right-padded batch, padding dropped on the host, work plan) at HEAD" 8 attn_ > /dev/null; echo "attn table rc=$?"
cat gpurun_out/${tag}_attn_pmc.md | cut -c1-260
rm -rf gpurun_out/${tag}_fetch/*/*agent_info.csv gpurun_out/${tag}_pad_*/*/*agent_info.csv
bash tools/r05_prof.sh ${tag}_padded --padded
bash tools/r05_prof.sh ${tag}_packed --packed --seq 8192 --batch 2 --n-dsus 2048
python bench.py --through-trainer --padded --steps 10 --warmup 3 > gpurun_out/${tag}_through_trainer_padded.json 2>/dev/null; echo "through-trainer padded rc=$?"
python bench.py --through-trainer --steps 10 --warmup 3 > gpurun_out/${tag}_through_trainer.json 2>/dev/null; echo "through-trainer rc=$?"
# the reference's default SFT geometry (2 rows x 2048, grad-accum 4) through the trainer: the window as one batch against the micro-batch loop
python bench.py --through-trainer --batch 2 --steps 10 --warmup 3 > gpurun_out/${tag}_through_trainer_b2.json 2>/dev/null; echo "through-trainer b2 rc=$?"
python bench.py --through-trainer --batch 2 --padded --steps 10 --warmup 3 > gpurun_out/${tag}_through_trainer_b2_padded.json 2>/dev/null; echo "through-trainer b2 padded rc=$?"
python bench.py --through-trainer --cpt --batch 16 --seq 768 --steps 10 --warmup 3 > gpurun_out/${tag}_through_trainer_cpt_b16_s768.json 2>/dev/null; echo "through-trainer cpt rc=$?"
python bench.py --through-trainer --cpt --batch 16 --seq 768 --padded --steps 10 --warmup 3 > gpurun_out/${tag}_through_trainer_cpt_b16_s768_padded.json 2>/dev/null; echo "through-trainer cpt padded rc=$?"
fi
