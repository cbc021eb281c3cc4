#!/bin/bash
# host-to-host rate of tgx_encode_batch_host by chunk size, caller buffers pageable and page-locked (bench.py's e2e leg)
out=${1:-gpurun_out/r02/e2e_sweep.txt}
: > $out
for mb in 64 128 256 512 2048; do
  echo "chunk=$mb MiB" >> $out
  TGX_E2E_CHUNK_MB=$mb timeout -k 10 200 python bench.py --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ', {k: v for k, v in d.items() if k.startswith('e2e')})" >> $out
done
cat $out
