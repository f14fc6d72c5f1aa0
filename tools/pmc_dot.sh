#!/bin/bash
# PMC counters of the dot-volume kernels, one rocprofv3 --pmc pass per counter group (no other trace domains):
#   tools/pmc_dot.sh <tag> <kernel 1|2|3> [B]
export TMPDIR=/tmp
TAG=$1; KERN=$2; B=${3:-32}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/pmc_dot_$TAG
mkdir -p $OUT
cd /tmp
i=0
for G in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
         "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM" "TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $OUT/g$i -o p -- python $ROOT/tools/perf_dot.py $KERN $B 8 64 4 > $OUT/g$i.log 2>&1
done
python - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "cv_dot" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
res = {k: {"per_launch": acc[k] / n[k], "launches": n[k]} for k in sorted(acc)}
json.dump(res, open("$OUT/summary.json", "w"), indent=1)
for k, v in res.items(): print(f"{k:36s} {v['per_launch']:16.1f}  ({v['launches']} launches)")
PY
