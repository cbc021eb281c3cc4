#!/bin/bash
# encode6_kernel by blocks per CU (TGX_E6_BPC) on the shapes of tools/e6_shapes.py
out=${1:-gpurun_out/r02/e6_bpc.txt}
: > $out
for b in 1 2; do
  echo "TGX_E6_BPC=$b" >> $out
  TGX_E6_BPC=$b timeout -k 10 400 python tools/e6_shapes.py 2>&1 | grep -E "thr=None|thr=2048 |thr=32768" >> $out || exit 1
done
