#!/bin/bash
# Compares the two traversals of the one-launch pipeline on the benchmark scenes (run on the GPU box).
out=gpurun_out/$1; mkdir -p $out
for t in walk stack; do
  for sc in "soup10000000 1920 1080 32 1 0 134217728" "teapot_lit 1920 1080 32 1 0 134217728" "rtiow 1920 1080 32 1 0 134217728" "cornell 1920 1080 64 1 0 134217728"; do
    set -- $sc
    RT_TRAVERSAL=$t timeout -k 10 240 python tools/quick_bench.py $sc > $out/${t}_$1.log 2>&1 || exit 1
    echo "$t $1: $(grep 'rep 2' $out/${t}_$1.log | sed 's/.*-> //; s/;.*//')"
  done
done
