#!/bin/bash
# The evidence of a round in one GPU call: PMC + kernel-trace profiles of every bench leg (tools/prof_legs.sh), the
# headline command under rocprofv3 --kernel-trace --stats, and the plain bench line.  usage: tools/round_profiles.sh OUTDIR
# Afterwards copy OUTDIR/r03_* into profiles/.
OUT=$1
mkdir -p "$OUT"
tools/prof_legs.sh "$OUT" cornell teapot teapot_lit spheres soup10m || exit 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# bench.py only accepts PMC profiles it finds under profiles/
cp "$OUT"/r03_pmc_*.json profiles/ || exit 1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_trace" -o trace -- python3 bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline > "$OUT/r03_bench_line_under_rocprof.json" 2> "$OUT/bench_trace.log" || { tail -n 5 "$OUT/bench_trace.log"; exit 1; }
cp "$(find "$OUT/bench_trace" -name '*kernel_stats.csv' | head -n 1)" "$OUT/r03_bench_kernel_stats.csv" || exit 1
# ... and quotes the kernel-trace averages it finds there
cp "$OUT"/r03_*_kernel_stats.csv profiles/ || exit 1
timeout -k 10 900 python3 bench.py > "$OUT/r03_bench_line.json" 2> "$OUT/bench.log" || { tail -n 5 "$OUT/bench.log"; exit 1; }
echo done
