#!/bin/bash
# Round 5: AdamW of every gradient bucket under the rest of the backward (SSI_ADAMW_OVERLAP=1) against AdamW behind the backward (=0), one GPU run
out=gpurun_out; mkdir -p $out
run() { tag=$1; shift; for v in 1 0 1 0; do SSI_ADAMW_OVERLAP=$v python bench.py --no-cpu-baseline --steps ${STEPS:-12} --warmup 4 "$@" 2>>$out/r05_ao_err.log | python -c "import json,sys; d=json.load(sys.stdin); print('$tag overlap=$v', round(d['value']), round(d['ms_per_step'],2), 'loss', d['config']['last_loss'])"; done; }
run headline
run b2_s2048 --batch 2 --seq 2048
run padded --padded
run headline_ga4 --grad-accum 4 --steps 5 --warmup 2
