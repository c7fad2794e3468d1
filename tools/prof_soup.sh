#!/bin/bash
# PMC passes (memory side) for the 10 M soup: RT_PIPE=<pipeline> tools/prof_soup.sh OUTDIR
OUT=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
i=0
for p in "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_READ_sum GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_READ_sum SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_BUSY_CYCLES"; do
  timeout -s KILL 150 rocprofv3 --pmc $p --output-format csv -d "$OUT/pass$i" -o pmc -- python3 tools/quick_bench.py soup10000000 1920 1080 32 1 ${RT_PIPE:-0} 134217728 > "$OUT/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -n 5 "$OUT/pass$i.log"; [ $i -eq 2 ] || exit 1; }
  i=$((i+1))
done
python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.txt"
grep "rep 2" "$OUT/pass0.log" | sed 's/.*rays \([0-9]*\) .*-> \([0-9.]*\) Mrays.*/rays \1 rate \2/' >> "$OUT/summary.txt"
