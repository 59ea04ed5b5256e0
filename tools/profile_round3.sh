#!/bin/bash
# The records of round 3 that are kept under profiles/ (run on the GPU box from the repository root):
#   tools/profile_round3.sh PART   ->  gpurun_out/...   (PART: headline | scenes | lines | shares | waves | cli)
# headline: counter passes + kernel trace of the meshDragon 4K frame; scenes: the same passes for the other configurations at 4K;
# lines: bench lines without counters; shares: what one GPU of 2 / 4 / 8 renders per frame; waves: wave lives of the headline
# frame and what the longest wave's life is made of; cli: the command line's phases and its frames in flight.
set -e
part=${1:?part}; root=$(pwd); out=$root/gpurun_out/round3
mkdir -p "$out"
case $part in
headline)
  bash tools/profile_headline.sh r3h > "$out/headline.log" 2>&1
  timeout -k 10 200 python3 bench.py > "$root/gpurun_out/bench_r3h.json" 2> "$out/bench_r3h.err"
  cut -c1-200 "$root/gpurun_out/bench_r3h.json" ;;
scenes)
  for pair in c2:scenes/lambertBPHSpheres.json c3:scenes/meshMedBunny.json c5a:scenes/meshTransparentBunny.json ico7:icosphere:7; do
    bash tools/profile_scene_all.sh "${pair%%:*}" "${pair#*:}" || exit 1
  done ;;
lines)
  : > "$out/bench_scenes.jsonl"
  for sc in scenes/sphere.json scenes/heliumAttenuatedSphere.json scenes/boxedSpheres.json scenes/meshTorus.json scenes/meshReflectedSquircle.json icosphere:6 icosphere:8; do
    timeout -k 10 200 python3 bench.py --scene "$sc" --no-cpu-baseline 2>> "$out/bench_scenes.log" | tail -1 >> "$out/bench_scenes.jsonl" || exit 1
  done
  python3 -c "
import json
for l in open('$out/bench_scenes.jsonl'): j = json.loads(l); print(j['config']['workload'], round(j['ms_per_step'], 4), 'ms', round(j['value']), 'Mrays/s', 'frac', round(j['roofline']['frac'], 3) if j['roofline'] else None)" ;;
shares)
  : > "$out/band_shares.jsonl"
  for n in 2 4 8; do
    RT_BENCH_SHARE_OF=$n timeout -k 10 200 python3 bench.py --steps 400 --warmup 20 --no-cpu-baseline 2>> "$out/bench_scenes.log" | tail -1 >> "$out/band_shares.jsonl" || exit 1
  done
  python3 -c "
import json
for l in open('$out/band_shares.jsonl'): j = json.loads(l); print(j['config']['partition'][:60], '| ms per frame', round(j['ms_per_step'], 4), '| one at a time', round(j['single_frame']['ms_per_frame_latency'], 4), j['value_is'])"
  timeout -k 10 300 python3 tools/exp_share_launch_only.py 2>&1 | grep "share of" > "$out/share_launch_only.txt"; cat "$out/share_launch_only.txt" ;;
waves)
  RT_AMD_LIB=build/variants/librt_amd_times.so timeout -k 10 200 python3 tools/wave_times.py meshDragon 3840x2160 2>&1 | grep -v amdgpu.ids > "$out/wave_lives.txt"
  RT_EXP_CLK=1 RT_AMD_TILE=8 RT_AMD_LIB=build/variants/librt_amd_timesclk.so timeout -k 10 200 python3 tools/exp_lone_wave_parts.py 2>&1 | grep "skip\|every" > "$out/lone_wave_parts.txt"
  RT_AMD_TILE=8 RT_AMD_LIB=build/variants/librt_amd_times.so timeout -k 10 200 python3 tools/exp_lone_wave.py 2>&1 | grep "tile row" > "$out/lone_wave_windows.txt"
  head -3 "$out/wave_lives.txt"; cat "$out/lone_wave_parts.txt" | cut -c1-200 ;;
cli)
  for rep in 1 2 3; do
    (cd assets && timeout -k 10 120 ../raytracer_amd/run -s scenes/meshDragon.json -w 3840 -h 2160 -i /tmp/dragon_4k.png --stats --frames 40 --counters ../profiles/r3_r3h_counters.json > "$out/cli_stats_dragon.json" 2> "$out/cli_stats_dragon.log")
  done
  (cd assets && timeout -k 10 120 ../raytracer_amd/run -s scenes/meshDragon.json -w 3840 -h 2160 -i /tmp/dragon_4k_g.png --stats --frames 40 --gpus 1 > "$out/cli_stats_dragon_gpus1.json" 2> "$out/cli_stats_dragon_gpus1.log")
  tail -1 "$out/cli_stats_dragon.json"; tail -1 "$out/cli_stats_dragon_gpus1.json" ;;
esac
