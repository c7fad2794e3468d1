#!/bin/bash
# Variant builds of the library (variants/librt_<name>.so, tools/build_variant.sh) against the built one (run on the
# GPU box): tools/variant_compare.sh OUTNAME "scene scene ..." [variant ...]; every variant when none is named.
out=gpurun_out/$1; mkdir -p $out
scenes=${2:-teapot teapot_lit}; shift 2
names=${@:-$(ls variants | sed -n 's/^librt_\(.*\)\.so$/\1/p')}
for v in base $names; do
  lib=""; [ $v != base ] && lib="$PWD/variants/librt_$v.so"
  for sc in $scenes; do
    spp=64; [ $sc = soup10000000 ] && spp=32
    RT_MI355X_LIB=$lib timeout -k 10 240 python tools/quick_bench.py $sc 1920 1080 $spp 0 0 134217728 > $out/${v}_$sc.log 2>&1 || { echo "$v $sc FAILED"; tail -n 3 $out/${v}_$sc.log; exit 1; }
    echo "$v $sc: $(grep 'rep 2' $out/${v}_$sc.log | sed 's/.*-> //; s/;.*//')"
  done
done
