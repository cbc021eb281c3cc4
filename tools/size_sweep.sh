#!/bin/bash
# bench.py by corpus size (kernel times per pass); extra arguments go to bench.py, e.g. --distinct-scores
out=$1; shift
: > $out
for mb in 10 64 256 512 1024; do
  python bench.py --no-e2e --no-cpu-baseline --size-mb $mb "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['bytes_per_gpu'], d['value'], d['ms_per_step'], d['kernel_ms_per_step'])" >> $out
done
cat $out
