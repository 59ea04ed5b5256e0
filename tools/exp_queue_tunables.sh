for cfg in "" "RT_AMD_OCTANT_FIRST_MIN=0" "RT_AMD_OCTANT_FIRST_MIN=4000000000" "RT_AMD_QUEUE_SMALL=0" "RT_AMD_QUEUE_SMALL=4000000000" "RT_AMD_OCTANT_FIRST_MIN=0 RT_AMD_QUEUE_SMALL=4000000000" "RT_AMD_OCTANT_FIRST_MIN=0 RT_AMD_QUEUE_SMALL=0"; do
  echo "== $cfg"
  env $cfg timeout -k 10 200 python tools/bench_scenes.py --sizes 512x512,1280x720,1920x1080,3840x2160 meshTransparentBunny 2>/dev/null | grep mesh || exit 1
done
