#!/bin/bash
# LANES against PATHS (stackless walk) on cache-resident and HBM scenes: tools/lanes_sweep2.sh OUTNAME
out=gpurun_out/$1; mkdir -p $out
run() { # label, scene, pipe, env...
  label=$1; scene=$2; pipe=$3; shift; shift; shift
  env "$@" timeout -k 10 200 python tools/quick_bench.py $scene 1920 1080 32 1 $pipe 134217728 > $out/$label.log 2>&1 || { echo "$label failed"; tail -3 $out/$label.log; return 1; }
  echo "$label: $(grep 'rep 2' $out/$label.log | sed 's/.*-> //; s/;.*//')"
}
for sc in teapot_lit soup1000000 soup100000; do
  run ${sc}_paths_walk $sc 7 || exit 1
  for b in 4 6 8; do
    run ${sc}_lanes_b$b $sc 9 RT_LANES_FIRST=lanes RT_LANES_WAVES=$(( b > 5 ? b : 5 )) RT_BLOCKS_PER_CU=$b || exit 1
    run ${sc}_lock_b$b $sc 9 RT_LANES_FIRST=lockstep RT_LANES_WAVES=$(( b > 5 ? b : 5 )) RT_BLOCKS_PER_CU=$b || exit 1
  done
done
