#!/bin/bash
# Kernel trace of one scene rendered through the library (run on the GPU box from the repository root):
#   tools/profile_scene.sh TAG SCENE WxH [reps]  ->  gpurun_out/prof_TAG/trace/ and gpurun_out/prof_TAG/kernel_stats.csv
# Shows how a frame's time is split among the launches of a scene that needs several (bounce rays from queues).
set -e
tag=${1:?tag}; scene=${2:?scene}; size=${3:-3840x2160}; reps=${4:-5}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 "$root/tools/bench_scenes.py" --sizes "$size" --reps "$reps" "$scene" > "$out/run.log" 2>&1
cd "$root"
cp "$(find "$out/trace" -name '*kernel_stats.csv' | head -1)" "$out/kernel_stats.csv"
tail -3 "$out/run.log"
cut -c1-200 "$out/kernel_stats.csv"
