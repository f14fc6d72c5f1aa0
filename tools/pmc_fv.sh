#!/bin/bash
# PMC counters of fv_mlp_k: tools/pmc_fv.sh <tag> [B]
export TMPDIR=/tmp
TAG=$1; B=${2:-32}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/pmc_fv_$TAG
mkdir -p $OUT
cd /tmp
i=0
for G in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU" \
         "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_IFETCH" "TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $OUT/g$i -o p -- python $ROOT/tools/perf_fv.py $B 7 64 3 > $OUT/g$i.log 2>&1
done
python - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fv_mlp_k" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
res = {k: {"per_launch": acc[k] / n[k], "launches": n[k]} for k in sorted(acc)}
json.dump(res, open("$OUT/summary.json", "w"), indent=1)
for k, v in res.items(): print(f"{k:36s} {v['per_launch']:18.1f}  ({v['launches']} launches)")
PY
