#!/bin/bash
# usage (on the GPU box, repo root): bash tools/r03_measure.sh <tag>
# through-trainer line, headline line on the same box, three PMC passes of the bench command -> per-kernel counter table.
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python bench.py --through-trainer --steps 20 --warmup 5 > gpurun_out/${tag}_through_trainer.json 2> gpurun_out/${tag}_through_trainer.err
echo "through-trainer rc=$?"; cut -c1-1500 gpurun_out/${tag}_through_trainer.json; tail -3 gpurun_out/${tag}_through_trainer.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_headline.json 2> gpurun_out/${tag}_headline.err
echo "headline rc=$?"; cut -c1-400 gpurun_out/${tag}_headline.json
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gemm-timing"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_fetch -- $CMD > gpurun_out/${tag}_fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_write -- $CMD > gpurun_out/${tag}_write.log 2>&1; echo "write rc=$?"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${tag}_sq -- $CMD > gpurun_out/${tag}_sq.log 2>&1; echo "sq rc=$?"
python tools/pmc_table.py gpurun_out/${tag}_fetch gpurun_out/${tag}_write gpurun_out/${tag}_sq gpurun_out/${tag}_pmc_kernels.md "${tag}: per-kernel counters of \`$CMD\` at HEAD" 14 > /dev/null; echo "table rc=$?"
python tools/pmc_table.py gpurun_out/${tag}_fetch gpurun_out/${tag}_write gpurun_out/${tag}_sq gpurun_out/${tag}_attn_pmc.md "${tag}: attention kernels, counters of \`$CMD\` at HEAD" 3 attn_ > /dev/null
cat gpurun_out/${tag}_pmc_kernels.md | cut -c1-260
# the big CSVs stay on the box
rm -rf gpurun_out/${tag}_fetch/*/*agent_info.csv
