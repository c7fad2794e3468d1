#!/bin/bash
# PATHS against the two LANES traversals over scene sizes (where AUTO's thresholds sit).
out=gpurun_out/rr3; mkdir -p $out
for n in 30000 100000 300000 600000 1000000; do
  sc="soup$n 1920 1080 32"
  timeout -k 10 240 python tools/quick_bench.py $sc 0 7 134217728 > $out/paths_$n.log 2>&1 || exit 1
  echo "soup$n paths: $(grep 'rep 2' $out/paths_$n.log | sed 's/.*-> //; s/;.*//')"
  for t in walk chain; do
    RT_LANES_TRAVERSAL=$t timeout -k 10 240 python tools/quick_bench.py $sc 0 9 134217728 > $out/${t}_$n.log 2>&1 || exit 1
    echo "soup$n lanes $t: $(grep 'rep 2' $out/${t}_$n.log | sed 's/.*-> //; s/;.*//')"
  done
done
