#!/bin/bash
# encode5_kernel geometry sweep on the bench workload: TGX_WAVES x TGX_BPC x TGX_PPL (kernel times from bench.py)
out=${1:-gpurun_out/r02/e5_sweep.txt}; shift
cfgs=${E5_CFGS:-"16 2 1;12 2 1;8 4 1;8 3 1;12 2 2;16 1 2;8 2 2"}
: > $out
IFS=';' read -ra arr <<< "$cfgs"
for cfg in "${arr[@]}"; do
  set -- $cfg
  echo "waves=$1 bpc=$2 ppl=$3" >> $out
  TGX_WAVES=$1 TGX_BPC=$2 TGX_PPL=$3 timeout -k 10 120 python bench.py --no-e2e --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ', d['kernel_ms_per_step'], d['value'])" >> $out
done
