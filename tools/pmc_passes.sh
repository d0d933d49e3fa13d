#!/bin/bash
# PMC passes over one layer's three entry points (developer tool; run on the GPU box from the repo root):
#   bash tools/pmc_passes.sh <out dir> [layer] [rows]
# One rocprofv3 run per counter set (the SQ block holds 8 counters, FETCH_SIZE and WRITE_SIZE do not fit together).
set -e
OUT=$1; LAYER=${2:-cfg2_hidden}; ROWS=${3:-16384}
ROOT=$(pwd)
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
B="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES"
for which in fwd bwd_x bwd_w; do
  timeout -s KILL 60 rocprofv3 --pmc $A --output-format csv -d "$ROOT/$OUT" -o ${which}_A -- python3 "$ROOT/tools/run_one.py" $which $ROWS $LAYER > /dev/null 2>&1
  timeout -s KILL 60 rocprofv3 --pmc $B --output-format csv -d "$ROOT/$OUT" -o ${which}_B -- python3 "$ROOT/tools/run_one.py" $which $ROWS $LAYER > /dev/null 2>&1
  timeout -s KILL 60 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$ROOT/$OUT" -o ${which}_FETCH -- python3 "$ROOT/tools/run_one.py" $which $ROWS $LAYER > /dev/null 2>&1
  timeout -s KILL 60 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$ROOT/$OUT" -o ${which}_WRITE -- python3 "$ROOT/tools/run_one.py" $which $ROWS $LAYER > /dev/null 2>&1
  echo "$which done"
done
