#!/bin/bash
# Instruction mix of one scene's frame (run on the GPU box from the repository root):
#   tools/profile_scene_pmc.sh TAG SCENE  ->  gpurun_out/prof_TAG/pmc/ and a summary on stdout
# One counter pass (eight SQ counters) over bench.py --scene SCENE at 3840x2160.
set -e
tag=${1:?tag}; scene=${2:?scene}
root=$(pwd); out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d "$out/pmc" -- python3 $root/bench.py --scene "$scene" --steps 5 --warmup 2 --no-cpu-baseline > "$out/pmc.log" 2>&1
cd "$root"
python3 tools/pmc_summary.py "$(dirname "$(find "$out/pmc" -name '*counter_collection.csv' | head -1)")"
tail -1 "$out/pmc.log" | cut -c1-300
