#!/bin/bash
# Launch shapes of the chain walk (block size, top records in LDS) on the HBM soups (run on the GPU box).
out=gpurun_out/$1; shift; mkdir -p $out
SCENES=${@:-soup10000000 soup1000000}
run() { # name, env...
  local name=$1; shift
  for sc in $SCENES; do
    env "$@" timeout -k 10 300 python tools/quick_bench.py $sc 1920 1080 32 0 0 134217728 > $out/${name}_$sc.log 2>&1 || { echo "$name $sc FAILED"; tail -n 5 $out/${name}_$sc.log; return 1; }
    echo "$name $sc: $(grep 'rep 2' $out/${name}_$sc.log | sed 's/.*-> //; s/;.*//')"
  done
}
run none RT_CHAIN_TOP_RECORDS=0 || exit 1
run 256x4_56 X=1 || exit 1
run 512x2_120 RT_CHAIN_THREADS=512 || exit 1
run 1024x1_248 RT_CHAIN_THREADS=1024 || exit 1
K8=$PWD/variants/librt_k8.so
run k8_256x4_184 RT_MI355X_LIB=$K8 || exit 1
run k8_512x2_376 RT_MI355X_LIB=$K8 RT_CHAIN_THREADS=512 || exit 1
run k8_1024x1_760 RT_MI355X_LIB=$K8 RT_CHAIN_THREADS=1024 || exit 1
echo done
