#!/bin/bash
# The records of a round that are kept under profiles/ (run on the GPU box from the repository root):
#   tools/profile_round.sh TAG  ->  gpurun_out/round_TAG/...
# headline counters and kernel trace (tools/profile_headline.sh), bench lines of the other configurations, the launches of a
# glass-bunny frame, the per-phase times of the command line on the dragon.
set -e
tag=${1:?tag}; root=$(pwd); out=$root/gpurun_out/round_$tag
mkdir -p "$out"
bash tools/profile_headline.sh "$tag" > "$out/headline.log" 2>&1
: > "$out/bench_scenes.jsonl"
for sc in scenes/meshTransparentBunny.json scenes/lambertBPHSpheres.json scenes/meshMedBunny.json icosphere:6 icosphere:7; do
  timeout -k 10 200 python3 bench.py --scene "$sc" --no-cpu-baseline 2>> "$out/bench_scenes.log" | tail -1 >> "$out/bench_scenes.jsonl"
done
# what one GPU of N does per frame (the bands of rank 0 of N, exchange with a world of one): frames in flight, and one at a time
: > "$out/band_shares.jsonl"
for n in 2 4 8; do
  RT_BENCH_SHARE_OF=$n timeout -k 10 200 python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>> "$out/bench_scenes.log" | tail -1 >> "$out/band_shares.jsonl"
done
bash tools/profile_scene.sh "${tag}_c5" meshTransparentBunny 3840x2160 3 > "$out/c5_profile.log" 2>&1
f=$(find "gpurun_out/prof_${tag}_c5/trace" -name '*kernel_trace.csv' | head -1)
python3 tools/frame_timeline.py "$f" > "$out/c5_frame_timeline.txt"
cp "gpurun_out/prof_${tag}_c5/kernel_stats.csv" "$out/c5_kernel_stats.csv"
# (three processes, the last one's phases are kept: the first process on a box pays for the machine's first use of the GPU)
for rep in 1 2 3; do
  (cd assets && timeout -k 10 120 ../raytracer_amd/run -s scenes/meshDragon.json -w 3840 -h 2160 -i /tmp/dragon_4k.png --stats > "$out/cli_stats_dragon.json" 2> "$out/cli_stats_dragon.log")
done
tail -2 "$out/c5_frame_timeline.txt"; cat "$out/cli_stats_dragon.json"
