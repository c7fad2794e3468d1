#!/bin/bash
# The built library with and without an environment setting (run on the GPU box):
# tools/env_compare.sh OUTNAME "scene scene ..." VAR=VALUE [VAR=VALUE ...]
out=gpurun_out/$1; mkdir -p $out
scenes=$2; shift 2
for v in base "$@"; do
  for sc in $scenes; do
    spp=64; [ $sc = soup10000000 ] && spp=32
    tag=$(echo $v | tr '= ' '__')
    if [ "$v" = base ]; then set_env="X=1"; else set_env="$v"; fi
    env $set_env timeout -k 10 240 python tools/quick_bench.py $sc 1920 1080 $spp 0 0 134217728 > $out/${tag}_$sc.log 2>&1 || { echo "$v $sc FAILED"; tail -n 3 $out/${tag}_$sc.log; exit 1; }
    echo "$v $sc: $(grep 'rep 2' $out/${tag}_$sc.log | sed 's/.*-> //; s/;.*//')"
  done
done
