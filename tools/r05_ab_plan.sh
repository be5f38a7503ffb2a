#!/bin/bash
# Round 5: the attention work plan against the plan-less path, in ONE GPU run (boxes differ by several percent): right-padded batches after the
# unpadding, BASELINE config E's packed rows, and the headline as a regression check.  Output: gpurun_out/r05_ab_*.json
set -o pipefail
out=gpurun_out; mkdir -p $out
B="python bench.py --no-cpu-baseline --steps ${STEPS:-10} --warmup 4"
for rep in 1 2; do
  $B --padded                                   > $out/r05_ab_padded_plan_$rep.json   2>$out/r05_ab_err.log || exit 1
  $B --padded --no-attn-plan                    > $out/r05_ab_padded_noplan_$rep.json 2>>$out/r05_ab_err.log || exit 1
done
$B --packed --seq 8192 --batch 2 --n-dsus 2048                > $out/r05_ab_packed_plan.json   2>>$out/r05_ab_err.log || exit 1
$B --packed --seq 8192 --batch 2 --n-dsus 2048 --no-attn-plan > $out/r05_ab_packed_noplan.json 2>>$out/r05_ab_err.log || exit 1
$B                                                             > $out/r05_ab_headline.json      2>>$out/r05_ab_err.log || exit 1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05_ab_*.json")):
    d = json.load(open(f))
    print(f"{f.split('r05_ab_')[1][:-5]:22s} {d['value']:10.0f} tok/s  {d['ms_per_step']:8.2f} ms  frac {d.get('mfma_roofline_frac_step', 0):.4f}  {d['attention_backward']}")
PY
