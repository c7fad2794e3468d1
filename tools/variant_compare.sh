#!/bin/bash
# Variant builds of the library (variants/librt_<name>.so, names in $VARIANTS) against the built one on the LANES legs (run on the GPU box).
out=gpurun_out/rr6; mkdir -p $out
for v in base $VARIANTS; do
  lib=""; [ $v != base ] && lib="$PWD/variants/librt_$v.so"
  for sc in "teapot 1920 1080 64" "teapot_lit 1920 1080 64" "soup10000000 1920 1080 32"; do
    set -- $sc
    RT_MI355X_LIB=$lib timeout -k 10 240 python tools/quick_bench.py $sc 0 0 134217728 > $out/${v}_$1.log 2>&1 || exit 1
    echo "$v $1: $(grep 'rep 2' $out/${v}_$1.log | sed 's/.*-> //; s/;.*//')"
  done
done
