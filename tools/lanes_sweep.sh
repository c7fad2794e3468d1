#!/bin/bash
# Sweep of the LANES pipeline's knobs on the 10 M soup (run on the GPU box): tools/lanes_sweep.sh OUTNAME
out=gpurun_out/$1; mkdir -p $out
run() { # label, env...
  label=$1; shift
  env "$@" timeout -k 10 200 python tools/quick_bench.py ${SCENE:-soup10000000} 1920 1080 32 1 ${PIPE:-9} 134217728 > $out/$label.log 2>&1 || { echo "$label failed"; tail -3 $out/$label.log; return 1; }
  echo "$label: $(grep 'rep 2' $out/$label.log | sed 's/.*-> //; s/;.*//')"
}
for b in 1 2 3 4; do
  run lock_b$b RT_LANES_FIRST=lockstep RT_LANES_WAVES=4 RT_BLOCKS_PER_CU=$b || exit 1
  run lanes_b$b RT_LANES_FIRST=lanes RT_LANES_WAVES=4 RT_BLOCKS_PER_CU=$b || exit 1
done
PIPE=7 run paths_b2 RT_BLOCKS_PER_CU=2
PIPE=7 run paths_b4 RT_BLOCKS_PER_CU=4
PIPE=7 run paths_b8 RT_BLOCKS_PER_CU=8
