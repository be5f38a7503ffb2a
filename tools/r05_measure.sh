#!/bin/bash
# usage (GPU box, repo root): bash tools/r05_measure.sh <tag> [headline]
# Round 5: the secondary workloads of one box in ONE file, gpurun_out/<tag>_secondary.json — every line with the attention backward kernels it
# ran (bench.py's "attention_backward": ssi_attn_last_dispatch) so that a reader sees which path a number came from.  With "headline" also the
# judged artefacts of the headline (tools/refresh_profiles.sh: bench line with cpu_baseline, rocprofv3 kernel stats, FETCH / WRITE passes).
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ "$2" = "headline" ]; then bash tools/refresh_profiles.sh $tag || exit 1; fi
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
