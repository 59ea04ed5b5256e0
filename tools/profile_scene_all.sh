#!/bin/bash
# The rocprofv3 passes of tools/profile_headline.sh for any scene bench.py can time (run on the GPU box from the repository root):
#   tools/profile_scene_all.sh TAG SCENE   ->  gpurun_out/prof_TAG/{trace,pmc_sq,pmc_fp,pmc_fetch,pmc_write}/, summary_TAG.txt, bench_TAG.json
# SCENE: scenes/NAME.json or icosphere:K.  One counter pass per run (the hardware holds 8 SQ and 4 TCC counters at a time),
# the kernel trace in a run of its own, the plain bench line last.
set -e
tag=${1:?tag}; scene=${2:?scene}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
B="python3 $root/bench.py --scene $scene --steps 5 --warmup 2 --no-cpu-baseline"
timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 $root/bench.py --scene $scene --steps 20 --warmup 3 --no-cpu-baseline > "$out/trace.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d "$out/pmc_sq" -- $B > "$out/pmc_sq.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 --output-format csv -d "$out/pmc_fp" -- $B > "$out/pmc_fp.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d "$out/pmc_fetch" -- $B > "$out/pmc_fetch.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum --output-format csv -d "$out/pmc_write" -- $B > "$out/pmc_write.log" 2>&1
cd "$root"
{
  for p in pmc_sq pmc_fp pmc_fetch pmc_write; do echo "## pass $p"; python3 tools/pmc_summary.py "$(dirname "$(find "$out/$p" -name '*counter_collection.csv' | head -1)")" 'render_|[ :]q_|unpack_'; done
  echo "## kernel trace"; cat $(find "$out/trace" -name '*kernel_stats.csv' | head -1)
} > "$root/gpurun_out/summary_$tag.txt"
timeout -k 10 200 python3 bench.py --scene $scene --no-cpu-baseline 2> "$out/bench.err" | tail -1 > "$root/gpurun_out/bench_$tag.json"
cut -c1-160 "$root/gpurun_out/bench_$tag.json"
