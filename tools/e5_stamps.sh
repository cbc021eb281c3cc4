#!/bin/bash
# s_memtime stamps per phase of encode5_kernel (TGX_STAMPS=1) for a few geometries; shares, not run times
out=${1:-gpurun_out/r02/e5_stamps.txt}
: > $out
for cfg in "12 2 1" "6 2 1" "16 1 2" "8 1 2" "8 1 4"; do
  set -- $cfg
  echo "waves=$1 bpc=$2 ppl=$3" >> $out
  TGX_STAMPS=1 TGX_WAVES=$1 TGX_BPC=$2 TGX_PPL=$3 timeout -k 10 120 python bench.py --no-e2e --no-cpu-baseline --steps 1 --warmup 0 2>&1 | grep -a "stamps\|kernel_ms" | cut -c1-400 >> $out
done
