#!/bin/bash
# HBM traffic of the bench kernels: FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md),
# --pmc with --kernel-trace only.  usage: tools/pmc_traffic.sh <outdir-under-gpurun_out> [commit-label]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp
for P in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/$P -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-e2e --no-distinct > $OUT/$P.log 2>&1 || { echo "$P failed: stopping"; exit 1; }
done
python3 - $OUT ${2:-?} <<'PY'
import csv, glob, collections, json, sys
out = sys.argv[1]
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for d in glob.glob(f"{out}/{c}/*/*_counter_collection.csv"):
        with open(d) as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != c: continue
                k = row["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]
                agg[k] += float(row["Counter_Value"]); cnt[k] += 1
    for k, v in agg.items():
        res[k][c + "_KB_per_launch"] = v / cnt[k]
# gfx950: FETCH_SIZE reports half of the bytes of wide coalesced reads (guide); WRITE_SIZE is exact for
# 16-B-per-lane stores, partial lines are counted as whole requests
for k, v in res.items():
    v["hbm_bytes_per_launch_corrected"] = 2 * 1024 * v.get("FETCH_SIZE_KB_per_launch", 0) + 1024 * v.get("WRITE_SIZE_KB_per_launch", 0)
kern = {k: v for k, v in res.items() if k.startswith(("encode", "trace", "compact", "scan"))}
# the E-step sub-record's pass (bench.py's `estep`): the fused kernel, its redo kernel and the small Z / order kernels
ekern = {k: v for k, v in res.items() if k.startswith(("estep", "piece", "snip", "cut"))}
doc = {"workload": "bench.py defaults (1 GiB mixed, 32 000-entry spec vocabulary over a 64 MiB slice: 9 652 score values), --steps 1 --warmup 1 --no-e2e",
       "commit": sys.argv[2] if len(sys.argv) > 2 else "?",
       "kernels": kern,
       "hbm_bytes_per_pass_corrected": sum(v["hbm_bytes_per_launch_corrected"] for v in kern.values()),
       "estep_kernels": ekern,
       "estep_hbm_bytes_per_pass_corrected": sum(v["hbm_bytes_per_launch_corrected"] for v in ekern.values())}
json.dump(doc, open(out + "/pmc_traffic.json", "w"), indent=1)
print(open(out + "/pmc_traffic.json").read())
PY
