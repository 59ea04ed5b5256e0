#!/bin/bash
# Two quick rocprofv3 counter passes over the headline benchmark (instruction mix, waits); run on the GPU box from the
# repository root.  Output: counters per dispatch of the render kernel.
set -e
root=$(pwd); out=$root/gpurun_out/prof_tmp; rm -rf $out; mkdir -p $out; export TMPDIR=/tmp; cd /tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d "$out/pmc_sq" -- python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline > "$out/pmc_sq.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_FLAT SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_SCA --output-format csv -d "$out/pmc_b" -- python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline > "$out/pmc_b.log" 2>&1
cd $root
for p in pmc_sq pmc_b; do python3 tools/pmc_summary.py "$(dirname "$(find "$out/$p" -name '*counter_collection.csv' | head -1)")"; done
