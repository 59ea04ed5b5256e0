#!/bin/bash
# One counter pass over the headline benchmark: bytes the L2 caches write out and fetch per frame (run on the GPU box):
#   tools/pmc_write.sh TAG  ->  gpurun_out/prof_TAG/pmc_write, pmc_fetch
set -e
tag=${1:-w}; root=$(pwd); out=$root/gpurun_out/prof_$tag
mkdir -p "$out"; export TMPDIR=/tmp; cd /tmp
B="python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum --output-format csv -d "$out/pmc_write" -- $B > "$out/pmc_write.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d "$out/pmc_fetch" -- $B > "$out/pmc_fetch.log" 2>&1
cd "$root"
for p in pmc_write pmc_fetch; do python3 tools/pmc_summary.py "$(dirname "$(find "$out/$p" -name '*counter_collection.csv' | head -1)")"; done
