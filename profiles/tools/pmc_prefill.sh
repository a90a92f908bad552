#!/bin/bash
# SQ counters of the prefill kernel, three separate --pmc passes (no tracing flags besides --kernel-trace).
# usage (on the GPU box, from the repo root):  bash profiles/tools/pmc_prefill.sh > gpurun_out/pmc_prefill.txt
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/pmc_pass$i
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_pass$i -o t --output-format csv -- python3 $ROOT/profiles/tools/one_shape.py "$@" > /tmp/pmc_pass$i.log 2>&1
  echo "# pass $i: --pmc $set"
  python3 $ROOT/profiles/tools/pmc_sum.py $(find /tmp/pmc_pass$i -name "*counter_collection.csv" | head -1) mfma_r
done
