#!/bin/bash
out=gpurun_out/$1; shift; mkdir -p $out
SCENES=${@:-teapot teapot_lit}
run() { # name, env...
  local name=$1; shift
  for sc in $SCENES; do
    env "$@" timeout -k 10 200 python tools/quick_bench.py $sc 1920 1080 64 0 0 134217728 > $out/${name}_$sc.log 2>&1 || { echo "$name $sc FAILED"; tail -n 5 $out/${name}_$sc.log; return 1; }
    echo "$name $sc: $(grep 'rep 2' $out/${name}_$sc.log | sed 's/.*-> //; s/;.*//')"
  done
}
run top51_512x3 RT_WALK_THREADS=512 || exit 1
run top78_768x2 RT_WALK_THREADS=768 || exit 1
run top51_512x3_cam RT_WALK_THREADS=512 RT_LANES_FIRST=lanes || exit 1
run top78_768x2_cam RT_WALK_THREADS=768 RT_LANES_FIRST=lanes || exit 1
run w8_top38_512x4 RT_WALK_THREADS=512 RT_LANES_WAVES=8 || exit 1
run w8_top78_1024x2 RT_WALK_THREADS=1024 RT_LANES_WAVES=8 || exit 1
run top64_768x2 RT_WALK_THREADS=768 RT_WALK_TOP_KB=64 || exit 1
run top32_512x3 RT_WALK_THREADS=512 RT_WALK_TOP_KB=32 || exit 1
echo done
