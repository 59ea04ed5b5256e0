#!/bin/bash
# Fewest triangles of a leaf for which the wave's bundle is asked first (RT_CULL_MIN_TRIS; variants built with
# tools/build_variant.sh minN -DRT_CULL_MIN_TRIS=N): headline bench and two smaller cases.  Run on the GPU box.
for v in "" min1 min5 min8; do
  lib=${v:+build/variants/librt_amd_$v.so}
  echo "== ${v:-default (3)}"
  for rep in 1 2; do RT_AMD_LIB=$lib timeout -k 10 200 python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('  dragon 4K', round(j['ms_per_step'],4))" || exit 1; done
  RT_AMD_LIB=$lib timeout -k 10 200 python3 tools/bench_scenes.py --sizes 1920x1080,3840x2160 meshMedBunny meshDragon 2>/dev/null | cut -c1-40,65-90 | tail -4 || exit 1
done
