#!/bin/bash
# One PMC pass: tools/prof_pmc1.sh OUTDIR "COUNTERS" -- <quick_bench args>
OUT=$1; CNT=$2; shift; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -s KILL 120 rocprofv3 --pmc $CNT --output-format csv -d "$OUT/pass0" -o pmc -- python3 tools/quick_bench.py "$@" > "$OUT/pass0.log" 2>&1 || { echo "pass failed"; tail -n 5 "$OUT/pass0.log"; exit 1; }
python3 tools/pmc_summary.py "$OUT" | head -n 12
