#!/bin/bash
# What one GPU of 8 / 4 does per frame (the bands of rank 0, exchange with a world of one) under a few settings: the
# six-waves-per-SIMD variant also for shares, more frames in flight.  Run on the GPU box.
run() { echo "== $1"; shift; for n in 8 4; do env RT_BENCH_SHARE_OF=$n "$@" timeout -k 10 200 python3 bench.py --steps 60 --warmup 6 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('  share of $n: ms per frame', round(j['ms_per_step'],4), '| one frame at a time', round(j['single_frame']['ms_per_frame_latency'],4), j['value_is'][:40])" || exit 1; done; }
run default RT_X=0
run "six waves per SIMD from 10000 tiles" RT_AMD_DENSE_MIN_TILES=10000
run "four frames in flight" RT_BENCH_IN_FLIGHT=4
run "two frames in flight" RT_BENCH_IN_FLIGHT=2
run "six waves, four in flight" RT_AMD_DENSE_MIN_TILES=10000 RT_BENCH_IN_FLIGHT=4
