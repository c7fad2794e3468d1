#!/bin/bash
# Walk table against chain records by scene size, for renders and closest-hit queries (run on the GPU box).
out=gpurun_out/$1; mkdir -p $out
for n in 30000 60000 100000 300000 600000 1000000; do
  for t in walk chain; do
    r=$(RT_LANES_TRAVERSAL=$t timeout -k 10 200 python tools/quick_bench.py soup$n 1920 1080 32 0 9 134217728 2>&1 | grep 'rep 2' | sed 's/.*-> //; s/;.*//')
    q=$(RT_LANES_TRAVERSAL=$t timeout -k 10 200 python tools/trace_bench.py soup$n 8 2>&1 | grep 'rep 2' | sed 's/.*-> //; s/;.*//')
    echo "soup$n $t: render $r   queries $q"
  done
done
