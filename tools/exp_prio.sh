#!/bin/bash
# Measurement helper: the default build against builds whose waves raise their issue priority as their walk grows long
# (tools/build_variant.sh prioK -DRT_WALK_PRIO=K): whole frames, one eighth of a frame (one GPU of eight), the glass bunny.
libs="raytracer_amd/librt_amd.so $@"
run() { RT_AMD_LIB=$lib timeout -k 10 100 python bench.py --steps 30 --warmup 5 --no-cpu-baseline "${@:2}" 2>/dev/null \
      | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$lib $1', round(j['ms_per_step'],4))" || exit 1; }
for round in 1 2; do for lib in $libs; do run "dragon 4K"; done; done
for lib in $libs; do
  run "dragon 1080p" --width 1920 --height 1080
  run "dragon 512" --width 512 --height 512
  RT_BENCH_SHARE_OF=8 RT_BENCH_ONE_STREAM=1 run "dragon 4K share-of-8, one frame at a time"
  RT_BENCH_SHARE_OF=4 RT_BENCH_ONE_STREAM=1 run "dragon 4K share-of-4, one frame at a time"
  run "glass bunny 4K" --scene scenes/meshTransparentBunny.json
  run "glass bunny 1080p" --scene scenes/meshTransparentBunny.json --width 1920 --height 1080
  run "glass bunny 720p" --scene scenes/meshTransparentBunny.json --width 1280 --height 720
  run "torus 1080p" --scene scenes/meshTorus.json --width 1920 --height 1080
done
