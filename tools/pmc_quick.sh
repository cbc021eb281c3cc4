#!/bin/bash
# SQ / TA counter groups for the encode pass of bench.py (one bench step per group; --pmc with --kernel-trace only).
# usage: tools/pmc_quick.sh <outdir-under-gpurun_out> [extra bench args]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp
i=0
for P in "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" \
         "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_INSTS_VALU" \
         "TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/g$i -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-e2e --no-distinct "$@" > $OUT/g$i.log 2>&1 || { echo "group $i failed: stopping"; break; }
done
python3 - $OUT <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
res = collections.defaultdict(dict)
for d in sorted(glob.glob(out + "/g*/*/*_counter_collection.csv")):
    agg, cnt = collections.defaultdict(float), collections.Counter()
    with open(d) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"].split("(")[0].split("::")[-1][:40]
            agg[(k, row["Counter_Name"])] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
    for (k, c), v in agg.items():
        res[k][c] = v / cnt[(k, c)]
with open(out + "/summary.txt", "w") as f:
    for k, v in res.items():
        if k.startswith(("encode", "trace", "estep", "pair", "compact", "count")):
            f.write(k + "\n")
            for c, x in sorted(v.items()):
                f.write(f"    {c:42s} {x:18.1f}\n")
print(open(out + "/summary.txt").read())
PY
