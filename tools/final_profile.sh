#!/bin/bash
# The round's records: rocprofv3 stats + PMC traffic of the default bench command, the bench variants, the
# prune / merge passes at 1 GiB / 32 K and 256 MiB / 500 K, one prune and one merge run, the shape sweeps.
# usage: tools/final_profile.sh <outdir-under-gpurun_out> [commit-label]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
./tools/profile_bench.sh $1/bench_profile ${2:-?} > $O/bench_profile.log 2>&1
python bench.py > $O/bench_default.json 2> $O/bench_default.err
for ml in 4096 1024 256; do python bench.py --max-sample-len $ml --no-cpu-baseline --no-e2e > $O/bench_max_sample_len_$ml.json 2>/dev/null; done
python bench.py --vocab 65536 --no-e2e > $O/bench_vocab_65536.json 2>/dev/null
python bench.py --kind ascii --no-e2e > $O/bench_ascii.json 2>/dev/null
python bench.py --distinct-scores --no-e2e > $O/bench_distinct_scores.json 2>/dev/null
python bench.py --vocab 65536 --distinct-scores --no-e2e --no-cpu-baseline > $O/bench_vocab_65536_distinct_scores.json 2>/dev/null
python bench.py --vocab-slice-mb 2 --no-e2e --no-cpu-baseline > $O/bench_vocab_2MiB_slice.json 2>/dev/null
python bench.py --max-token-length 24 --distinct-scores --no-cpu-baseline --no-e2e > $O/bench_max_token_24_distinct_scores.json 2>/dev/null
python tools/merged_vocab_bench.py > $O/merged_vocab_bench.json 2>/dev/null
for sz in 10 64 256 512; do python bench.py --size-mb $sz --no-cpu-baseline --no-e2e | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($sz, d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"; done > $O/bench_by_size_spec_vocabulary.txt 2>/dev/null
for sz in 10 64 256 512; do python bench.py --size-mb $sz --distinct-scores --no-cpu-baseline --no-e2e | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($sz, d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"; done > $O/bench_by_size_distinct_scores.txt 2>/dev/null
python tools/generate_bench.py > $O/generate_bench.txt 2>&1
python tools/python_api_rate.py > $O/python_api_rate.json 2>/dev/null
python bench.py --size-mb 4096 --no-cpu-baseline --no-e2e --steps 3 --warmup 1 > $O/bench_4GiB.json 2>/dev/null
python bench.py --max-token-length 24 --no-cpu-baseline --no-e2e > $O/bench_max_token_24.json 2>/dev/null
python tests/measure/passes_bench.py 1024 32000 16 > $O/passes_1GiB.json 2>/dev/null
python tests/measure/passes_bench.py 256 32000 16 > $O/passes_256MiB.json 2>/dev/null
python tests/measure/passes_bench.py 256 500000 8 > $O/passes_256MiB_500k_vocab.json 2>/dev/null
python tools/prune_bench.py 256 500000 375000 > $O/prune_256MiB_500k_vocab.json 2> $O/prune_500k.err
python tools/prune_bench.py 256 32000 16000 > $O/prune_256MiB.json 2>/dev/null
python tools/merge_bench.py 256 32000 300 100 16 > $O/merge_256MiB.json 2>/dev/null
python tools/e6_shapes.py > $O/e6_shapes.txt 2>&1
TGX_KNOBS=1 python tools/e7_derived.py 256 short > $O/estep_second_subiteration_500k.txt 2>&1
TGX_KNOBS=1 TGX_HOST_TIMES=1 python tools/prune_bench.py 256 500000 375000 > /dev/null 2> $O/prune_500k_host_phases.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
python tests/measure/cpu_port_threads.py > $O/config0_cpu_port_threads.json 2>/dev/null
ls -la $O
