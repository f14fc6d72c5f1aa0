#!/bin/bash
# HBM-traffic counters for the bench workload (separate --pmc passes, guide §HBM):
#   tools/pmc_bench.sh <tag> [bench args...]
export TMPDIR=/tmp
TAG=$1; shift
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -o p -- python $ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" > $OUT/$C.log 2>&1
done
python $ROOT/tools/pmc_summarize.py $OUT; exit 0
python - <<PY
import csv, glob, collections, json
res = collections.defaultdict(lambda: {"calls": 0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob("$OUT/%s/**/*counter_collection.csv" % c, recursive=True)
    if not fs: continue
    n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] != c: continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")
        res[k][c] += float(r["Counter_Value"]); n[k] += 1
    for k, v in n.items(): res[k]["calls"] = v
rows = []
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["FETCH_SIZE"]):
    rows.append({"kernel": k, "calls": v["calls"], "fetch_KiB_per_call": v["FETCH_SIZE"] / max(v["calls"], 1), "write_KiB_per_call": v["WRITE_SIZE"] / max(v["calls"], 1),
                 "fetch_KiB_total": v["FETCH_SIZE"], "write_KiB_total": v["WRITE_SIZE"]})
json.dump(rows, open("$OUT/summary.json", "w"), indent=1)
for r in rows[:14]:
    print(f"{r['kernel'][:48]:48s} calls={r['calls']:5d} fetch/call={r['fetch_KiB_per_call']:10.1f} KiB  write/call={r['write_KiB_per_call']:10.1f} KiB")
PY
