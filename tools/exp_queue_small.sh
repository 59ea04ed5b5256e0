#!/bin/bash
# Measurement helper: the glass bunny with other thresholds for "16 rays per wave" (RT_AMD_QUEUE_SMALL) and, at 4K, with queues (and
# therefore launch grids) sized for fewer pixels than the frame has (RT_AMD_QUEUE_CAP_PIXELS).
S=${1:-scenes/meshTransparentBunny.json}
run() { timeout -k 10 100 python bench.py --scene $S --width ${1%x*} --height ${1#*x} --steps 30 --warmup 4 --no-cpu-baseline 2>/dev/null \
      | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1 $2', round(j['ms_per_step'],4))" || exit 1; }
for cap in 0 4000000 2000000 1000000; do
  if [ $cap = 0 ]; then unset RT_AMD_QUEUE_CAP_PIXELS; else export RT_AMD_QUEUE_CAP_PIXELS=$cap; fi
  run 3840x2160 "cap_pixels $cap"
done
unset RT_AMD_QUEUE_CAP_PIXELS
for sz in 3840x2160 2560x1440 1920x1080 1280x720 512x512; do
  for small in 131072 200000 300000 500000 1000000 2000000; do
    RT_AMD_QUEUE_SMALL=$small run $sz "small_below $small"
  done
done
