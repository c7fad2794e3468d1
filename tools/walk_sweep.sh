#!/bin/bash
# Launch shapes of the lanes kernel over the walk table (block size, blocks per CU, treetop bytes in LDS) on the
# L2-resident legs (run on the GPU box): tools/walk_sweep.sh OUTNAME [scene ...]
out=gpurun_out/$1; shift; mkdir -p $out
SCENES=${@:-teapot teapot_lit}
run() { # name, env...
  local name=$1; shift
  for sc in $SCENES; do
    env "$@" timeout -k 10 200 python tools/quick_bench.py $sc 1920 1080 64 0 0 134217728 > $out/${name}_$sc.log 2>&1 || { echo "$name $sc FAILED"; tail -n 5 $out/${name}_$sc.log; return 1; }
    echo "$name $sc: $(grep 'rep 2' $out/${name}_$sc.log | sed 's/.*-> //; s/;.*//')"
  done
}
[ -f variants/librt_r2.so ] && { run r2 RT_MI355X_LIB=$PWD/variants/librt_r2.so || exit 1; }
run top0_256x6 RT_WALK_TOP_KB=0 || exit 1
run top24_256x6 X=1 || exit 1
run top38_256x4 RT_WALK_BLOCKS_PER_CU=4 || exit 1
run top51_512x3 RT_WALK_THREADS=512 || exit 1
run top78_512x2 RT_WALK_THREADS=512 RT_WALK_BLOCKS_PER_CU=2 || exit 1
run top64_1024x1 RT_WALK_THREADS=1024 RT_WALK_TOP_KB=64 || exit 1
run top128_1024x1 RT_WALK_THREADS=1024 RT_WALK_TOP_KB=128 || exit 1
run top150_1024x1 RT_WALK_THREADS=1024 || exit 1
echo done
