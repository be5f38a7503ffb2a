#!/bin/bash
# Round 5: K split of the last partial round only (SSI_TAIL_SPLIT=1, default) against the split of the whole grid (=0), right-padded batches, one GPU run
out=gpurun_out; mkdir -p $out
B="python bench.py --no-cpu-baseline --steps ${STEPS:-10} --warmup 4 --padded"
for rep in 1 2; do
  SSI_TAIL_SPLIT=1 $B > $out/r05_tail_on_$rep.json  2>$out/r05_tail_err.log || exit 1
  SSI_TAIL_SPLIT=0 $B > $out/r05_tail_off_$rep.json 2>>$out/r05_tail_err.log || exit 1
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05_tail_*.json")):
    d = json.load(open(f))
    print(f"{f.split('r05_tail_')[1][:-5]:10s} {d['value']:10.0f} tok/s  {d['ms_per_step']:8.2f} ms  frac {d.get('mfma_roofline_frac_step', 0):.4f}")
PY
