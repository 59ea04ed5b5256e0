#!/bin/bash
# Measurement helper: the glass bunny with different orderings of the queued rays (direction grid per cube face x origin cell bits).
for cfg in "RT_AMD_SORT_DIRGRID=0" "RT_AMD_SORT_DIRGRID=4 RT_AMD_SORT_CELLBITS=4" "RT_AMD_SORT_DIRGRID=6 RT_AMD_SORT_CELLBITS=4" "RT_AMD_SORT_DIRGRID=4 RT_AMD_SORT_CELLBITS=3" "RT_AMD_SORT_DIRGRID=4 RT_AMD_SORT_CELLBITS=4 RT_AMD_QUEUE_SMALL=300000" "RT_AMD_SORT_DIRGRID=4 RT_AMD_SORT_CELLBITS=4 RT_AMD_QUEUE_SMALL=60000"; do
  echo "== $cfg"
  env $cfg timeout -k 10 200 python tools/bench_scenes.py --sizes ${1:-512x512,1280x720,1920x1080,3840x2160} meshTransparentBunny 2>/dev/null | grep mesh || exit 1
done
