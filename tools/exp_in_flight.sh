#!/bin/bash
# What one GPU of 8 / 4 renders per frame (the bands of rank 0, exchange with a world of one) with more frames in flight than the
# process has hardware queues by default (4): GPU_MAX_HW_QUEUES raised, RT_BENCH_IN_FLIGHT frames.  Run on the GPU box.
run() { q=$1; f=$2; n=$3; GPU_MAX_HW_QUEUES=$q RT_BENCH_IN_FLIGHT=$f RT_BENCH_SHARE_OF=$n timeout -k 10 200 python3 bench.py --steps 400 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('queues $q, $f in flight, share of $n: ms per frame', round(j['ms_per_step'],4), '| one frame at a time', round(j['single_frame']['ms_per_frame_latency'],4))" || exit 1; }
for n in 8 4; do
  for q in 4 8; do for f in 2 3 4 5 6; do run $q $f $n; done; done
done
