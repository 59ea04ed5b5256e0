#!/bin/bash
# How wide a bundle of rays may be before the walk stops putting it in front of a leaf's triangles (wave_bundle in
# rt_kernel_wave.hip): kernel times of the scenes with incoherent rays under several limits.  Run on the GPU box.
for cfg in "1.0 0.25" "0.25 0.05" "0.5 0.1" "2.0 0.5" "8.0 1.0" "0.1 0.02"; do
  set -- $cfg
  echo "== r0 <= $1, ksp <= $2"
  RT_AMD_CULL_R0=$1 RT_AMD_CULL_KSP=$2 timeout -k 10 200 python3 tools/bench_scenes.py --sizes 1920x1080,3840x2160 meshTransparentBunny meshReflectedSquircle meshTorus 2>/dev/null | cut -c1-40,65-90 || exit 1
done
