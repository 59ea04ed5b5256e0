#!/bin/bash
# Scalar data cache and instruction cache behaviour of the headline benchmark (one rocprofv3 counter pass each); run on
# the GPU box from the repository root.
set -e
root=$(pwd); out=$root/gpurun_out/prof_caches; rm -rf $out; mkdir -p $out; export TMPDIR=/tmp; cd /tmp
B="python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
timeout -s KILL 120 rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_TC_DATA_READ_REQ SQC_TC_STALL --output-format csv -d "$out/d" -- $B > "$out/d.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ --output-format csv -d "$out/i" -- $B > "$out/i.log" 2>&1
cd $root
for p in d i; do python3 tools/pmc_summary.py "$(dirname "$(find "$out/$p" -name '*counter_collection.csv' | head -1)")"; done
