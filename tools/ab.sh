#!/bin/bash
# A/B of library builds on one box: tools/ab.sh LIB_A LIB_B [bench.py arguments]; each library is run three times, alternating.
a=$1; b=$2; shift 2
for i in 1 2 3; do
  for lib in "$a" "$b"; do
    RT_AMD_LIB=$lib timeout -k 10 100 python bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$lib', round(j['ms_per_step'],4), 'ms', round(j['value'],1))"
  done
done
