#!/bin/bash
# SQ / TCC counter groups of any script's kernels, per kernel launch in launch order (not averaged).
# usage: tools/pmc_script.sh <outdir-under-gpurun_out> <script.py> [args]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp
i=0
for P in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
         "TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/g$i -- python3 $R/"$@" > $OUT/g$i.log 2>&1 || { echo "group $i failed: stopping"; break; }
done
python3 - $OUT <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
rows = collections.OrderedDict()
for d in sorted(glob.glob(out + "/g*/*/*_counter_collection.csv")):
    with open(d) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"].split("(")[0].split("::")[-1][:48]
            if not k.startswith(("estep7", "encode5", "trace", "mark", "emit")): continue
            rows.setdefault((int(row["Dispatch_Id"]), k), {})[row["Counter_Name"]] = float(row["Counter_Value"])
with open(out + "/per_launch.txt", "w") as f:
    for (disp, k), v in sorted(rows.items()):
        f.write(f"{disp:5d} {k}\n")
        for c, x in sorted(v.items()):
            f.write(f"        {c:32s} {x:18.0f}\n")
print(open(out + "/per_launch.txt").read())
PY
