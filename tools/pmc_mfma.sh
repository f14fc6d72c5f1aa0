#!/bin/bash
# MFMA utilisation per kernel from PMC counters: tools/pmc_mfma.sh <tag> [bench args]
export TMPDIR=/tmp
TAG=$1; shift
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/pmc_mfma_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -o p -- python $ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" > $OUT/sq.log 2>&1
python - <<PY
import csv, glob, collections, json, re
fs = glob.glob("$OUT/sq/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"[A-Za-z_0-9:]+(<[0-9a-z, ]+>)?", name); return m.group(0) if m else name[:40]
for r in csv.DictReader(open(fs[0])):
    k = short(r["Kernel_Name"]); acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": n[k] += 1
rows = []
for k, c in acc.items():
    if c.get("SQ_INSTS_MFMA", 0) <= 0: continue
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0            # summed over the 8 XCDs
    util = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc)   # 256 CUs x 4 SIMDs
    rows.append({"kernel": k, "launches": n[k], "mfma_insts": c["SQ_INSTS_MFMA"], "mfma_busy_cycles": c["SQ_VALU_MFMA_BUSY_CYCLES"],
                 "gpu_cycles": cyc, "mfma_util": util})
rows.sort(key=lambda r: -r["mfma_busy_cycles"])
json.dump(rows, open("$OUT/summary.json", "w"), indent=1)
for r in rows: print(f"{r['kernel'][:40]:40s} launches={r['launches']:5d} MFMA util={100*r['mfma_util']:5.1f}%  insts={r['mfma_insts']:.3g}")
PY
