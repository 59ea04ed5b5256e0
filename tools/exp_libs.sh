#!/bin/bash
# Measurement helper: tools/exp_libs.sh "SIZES" SCENE LIB...  - bench_scenes.py of one scene with each library build in turn.
sizes=$1; scene=$2; shift 2
for lib in "$@"; do
  echo "== $lib"
  RT_AMD_LIB=$lib timeout -k 10 200 python tools/bench_scenes.py --sizes "$sizes" "$scene" 2>/dev/null | grep -v "^scene" || exit 1
done
