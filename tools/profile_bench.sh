#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench command, then the PMC traffic passes (FETCH_SIZE and
# WRITE_SIZE in separate runs).  usage: tools/profile_bench.sh <outdir-under-gpurun_out> [commit-label]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --no-cpu-baseline --no-e2e > $OUT/stats_bench.json 2> $OUT/stats.err || { echo "stats run failed"; exit 1; }
cp $OUT/stats/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
cat $OUT/kernel_stats.csv | head -12
cd $R && ./tools/pmc_traffic.sh $1/traffic ${2:-?}
