#!/bin/bash
# usage (GPU box): tools/archive/lab/pmc_lab.sh <out-tag> <program and args...>   -- two PMC passes over one program, summary to stdout
# (the program itself goes after `--`: no env/bash wrapper under rocprofv3)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
tag=$1; shift
out=gpurun_out/$tag
mkdir -p "$out"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM"; do
  i=$((i + 1))
  rocprofv3 --pmc $set --output-format csv -d "$out/pmc$i" -o run -- "$@" > "$out/pmc$i.log" 2>&1 || { echo "rocprofv3 pass $i failed"; tail -5 "$out/pmc$i.log"; }
done
python tools/pmc_summary.py $(find "$out" -name "*counter_collection.csv") 2>&1 | grep -v "fill_kernel" | tee "$out/summary.txt" | cut -c1-900
