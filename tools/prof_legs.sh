#!/bin/bash
# rocprofv3 evidence for the bench legs (run on the GPU box): per leg one --kernel-trace --stats run and four --pmc
# passes over `bench.py --profile-leg LEG` (three identical launches of the leg's reference workload), then
# tools/make_pmc_json.py turns them into profiles/r03_pmc_<leg>.json + profiles/r03_<leg>_kernel_stats.csv.
# usage: tools/prof_legs.sh OUTDIR [LEG ...]
OUT=$1; shift
LEGS=${@:-cornell teapot teapot_lit spheres soup10m}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PASSES=(
 "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE"
 "FETCH_SIZE TCC_HIT_sum"
 "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum"
 "TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU"
)
for leg in $LEGS; do
  d="$OUT/$leg"; mkdir -p "$d"
  timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$d/trace" -o trace -- python3 bench.py --profile-leg $leg > "$d/trace.log" 2>&1 || { echo "$leg: trace run failed"; tail -n 5 "$d/trace.log"; exit 1; }
  i=0
  for p in "${PASSES[@]}"; do
    timeout -s KILL 240 rocprofv3 --pmc $p --output-format csv -d "$d/pass$i" -o pmc -- python3 bench.py --profile-leg $leg > "$d/pass$i.log" 2>&1 || { echo "$leg: pass $i failed"; tail -n 5 "$d/pass$i.log"; exit 1; }
    i=$((i+1))
  done
  python3 tools/make_pmc_json.py "$d" "$OUT/r03_pmc_$leg.json" "$OUT/r03_${leg}_kernel_stats.csv" || exit 1
done
echo done
