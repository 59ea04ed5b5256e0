#!/bin/bash
# Per-kernel times of a bench run of one scene (run on the GPU box from the repository root):
#   tools/trace_scene.sh scenes/meshTransparentBunny.json [rows]
# prints calls, average ns and share of the `rows` longest-running kernels (rocprofv3 --kernel-trace --stats), then two bench lines.
set -e
scene=${1:?scene}; rows=${2:-14}
root=$(pwd)
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/trace_scene
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trace_scene -- python3 $root/bench.py --scene $scene --no-cpu-baseline --steps 10 --warmup 2 > /tmp/trace_scene.log 2>&1
python3 - "$rows" <<'PY'
import csv, glob, sys
f = glob.glob('/tmp/trace_scene/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.reader(open(f)))[1:1 + int(sys.argv[1])]:
    print(r[0][:64].ljust(64), r[1].rjust(5), ("%.0f" % float(r[3])).rjust(9), r[4])
PY
cd "$root"
for i in 1 2; do timeout -k 10 120 python3 bench.py --scene $scene --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', round(d['ms_per_step'],4))"; done
