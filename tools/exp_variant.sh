#!/bin/bash
# Measures library variants (tools/build_variant.sh NAME and NAMEtimes) against each other: headline frame, one GPU's share of 8,
# glass bunny, the longest waves of two tile rows on their own, and the parity tests with the variant.
#   tools/exp_variant.sh NAME [NAME ...]      ("base" = the library in the tree and build/variants/librt_amd_times.so)
mkdir -p gpurun_out/variant
for v in "$@"; do
  if [ $v = base ]; then unset RT_AMD_LIB; T=build/variants/librt_amd_times.so; else export RT_AMD_LIB=build/variants/librt_amd_$v.so; T=build/variants/librt_amd_${v}times.so; fi
  echo "== $v"
  timeout -k 10 200 python3 bench.py --no-cpu-baseline 2>gpurun_out/variant/bench_$v.err | tail -1 | cut -c1-160 || exit 1
  RT_BENCH_SHARE_OF=8 timeout -k 10 200 python3 bench.py --steps 400 --warmup 20 --no-cpu-baseline 2>>gpurun_out/variant/bench_$v.err | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('share of 8', j['ms_per_step'], j['single_frame']['ms_per_frame_latency'])" || exit 1
  for sc in scenes/meshTransparentBunny.json scenes/meshMedBunny.json icosphere:7; do
    timeout -k 10 200 python3 bench.py --scene $sc --no-cpu-baseline 2>>gpurun_out/variant/bench_$v.err | tail -1 | cut -c1-160 || exit 1
  done
  RT_AMD_TILE=8 RT_AMD_LIB=$T timeout -k 10 200 python3 tools/exp_lone_wave.py 2>&1 | grep "tile row" | head -2 || exit 1
  if [ $v != base ]; then timeout -k 10 600 python3 -u -m pytest tests/test_gpu_parity.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -2 || exit 1; fi
done
