#!/bin/bash
# Headline frame under a few settings next to each other (run on the GPU box): five instead of six waves per SIMD, narrower
# and wider limits for the bundles that are put in front of a leaf's triangles, no bundle test at all.
run() { echo "== $1"; shift; for rep in 1 2; do env "$@" timeout -k 10 200 python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('  ms', round(j['ms_per_step'],4))" || exit 1; done; }
run default RT_X=0
run "five waves per SIMD" RT_AMD_DENSE_MIN_TILES=999999999
run "bundles r0 <= 0.3, ksp <= 0.05" RT_AMD_CULL_R0=0.3 RT_AMD_CULL_KSP=0.05
run "bundles r0 <= 3, ksp <= 0.5" RT_AMD_CULL_R0=3 RT_AMD_CULL_KSP=0.5
run "every triangle" RT_AMD_NO_CULL=1
