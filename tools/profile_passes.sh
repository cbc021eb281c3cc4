#!/bin/bash
# rocprofv3 kernel stats + PMC groups of the prune / merge corpus passes (tests/measure/passes_bench.py).
# usage: tools/profile_passes.sh <outdir-under-gpurun_out> <MiB> <vocab>
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; MIB=${2:-1024}; V=${3:-32000}
mkdir -p $OUT
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/tests/measure/passes_bench.py $MIB $V 16 > $OUT/passes.json 2> $OUT/stats.err || { echo "stats run failed"; tail -5 $OUT/stats.err; exit 1; }
cp $OUT/stats/*/*kernel_stats.csv $OUT/kernel_stats.csv
head -14 $OUT/kernel_stats.csv
i=0
for P in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
         "SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_WAVES" \
         "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
         "TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/g$i -- python3 $R/tests/measure/passes_bench.py $MIB $V 4 > $OUT/g$i.log 2>&1 || { echo "group $i failed: stopping"; break; }
done
python3 - $OUT <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
res = collections.defaultdict(dict)
for d in sorted(glob.glob(out + "/g*/*/*_counter_collection.csv")):
    agg, cnt = collections.defaultdict(float), collections.Counter()
    with open(d) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"].split("(")[0].split("::")[-1][:34]
            agg[(k, row["Counter_Name"])] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
    for (k, c), v in agg.items():
        res[k][c] = v / cnt[(k, c)]
with open(out + "/pmc_summary.txt", "w") as f:
    for k, v in res.items():
        if k.startswith(("encode", "trace", "estep", "pair")):
            f.write(k + "   (mean per launch)\n")
            for c, x in sorted(v.items()):
                f.write(f"    {c:42s} {x:18.1f}\n")
print(open(out + "/pmc_summary.txt").read())
PY
