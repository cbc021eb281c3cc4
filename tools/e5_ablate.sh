#!/bin/bash
# encode5_kernel with phases switched off (TGX_DEBUG=1 TGX_FLAGS: 1 no walk, 2 no relax, 8 every gather from slot 0/1):
# results are WRONG, only the kernel times mean something
out=${1:-gpurun_out/r02/e5_ablate.txt}
: > $out
for cfg in "0 1" "1 1" "2 1" "8 1" "10 1" "0 2" "1 2" "2 2" "8 2" "10 2"; do
  set -- $cfg
  echo "flags=$1 ppl=$2" >> $out
  TGX_DEBUG=1 TGX_FLAGS=$1 TGX_PPL=$2 timeout -k 10 120 python bench.py --no-e2e --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ', d['kernel_ms_per_step'])" >> $out
done
