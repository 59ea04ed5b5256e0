#!/bin/bash
# The headline benchmark without the tracer, under rocprofv3 --kernel-trace --stats, and without it again, on the same
# box (run on the GPU box from the repository root): the tracer's average kernel duration beside bench.py's own.
set -e
root=$(pwd); out=$root/gpurun_out/prof_tb; rm -rf $out; mkdir -p $out; export TMPDIR=/tmp
python3 bench.py --no-cpu-baseline | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('bench', j['ms_per_step'], j['roofline']['kernel_ms'])"
cd /tmp
timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline > "$out/trace.log" 2>&1
grep -o '"ms_per_step": [0-9.]*\|"kernel_ms": [0-9.]*' "$out/trace.log" | tr '\n' ' '; echo
cd $root
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_tb/trace/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'render_wave' in r['Name']: print('trace', r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
python3 bench.py --no-cpu-baseline | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('bench', j['ms_per_step'], j['roofline']['kernel_ms'])"
