#!/bin/bash
# LANES pipeline against PATHS on the scenes read through the caches / from HBM (run on the GPU box).
out=gpurun_out/$1; mkdir -p $out
for sc in "soup10000000 1920 1080 32 1" "teapot_lit 1920 1080 32 1" "soup1000000 1920 1080 32 1"; do
  set -- $sc
  timeout -k 10 240 python tools/quick_bench.py $sc 7 134217728 > $out/paths_$1.log 2>&1 || exit 1
  echo "paths $1: $(grep 'rep 2' $out/paths_$1.log | sed 's/.*-> //; s/;.*//')"
  for w in 4 6 8; do
    RT_LANES_WAVES=$w timeout -k 10 240 python tools/quick_bench.py $sc 9 134217728 > $out/lanes${w}_$1.log 2>&1 || exit 1
    echo "lanes w$w $1: $(grep 'rep 2' $out/lanes${w}_$1.log | sed 's/.*-> //; s/;.*//')"
  done
done
