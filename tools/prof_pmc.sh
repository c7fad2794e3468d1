#!/bin/bash
# Collects PMC counters for one quick_bench invocation, one rocprofv3 pass per counter group.
# usage: tools/prof_pmc.sh OUTDIR -- <quick_bench args>
set -u
OUT=$1; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
PASSES=(
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"
 "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
 "FETCH_SIZE TCC_HIT_sum"
 "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum"
 "GRBM_GUI_ACTIVE GRBM_COUNT SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAVES_EQ_64 SQ_INSTS_SMEM SQ_INSTS_FLAT"
)
i=0
for p in "${PASSES[@]}"; do
  timeout -s KILL 120 rocprofv3 --pmc $p --output-format csv -d "$OUT/pass$i" -o pmc -- python3 tools/quick_bench.py "$@" > "$OUT/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -n 5 "$OUT/pass$i.log"; exit 1; }
  i=$((i+1))
done
echo done
