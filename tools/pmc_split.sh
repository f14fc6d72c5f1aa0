#!/bin/bash
# SQ counters for the split-bf16 conv kernel on one layer shape: tools/pmc_split.sh <tag>
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/pmc_split_$1; mkdir -p $OUT; cd /tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $OUT/a -o p -- python $ROOT/tools/perf_split.py 32 one > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -o p -- python $ROOT/tools/perf_split.py 32 one > $OUT/b.log 2>&1
python - <<PY
import csv, glob, collections
for sub in "ab":
    fs = glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True)
    if not fs: print("no csv for", sub); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "conv3x3" not in k: continue
        k = "split" if "split" in k else "fp32lds"
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, c in acc.items():
        g = c["GRBM_GUI_ACTIVE"] / 8
        print(k, " ".join(f"{name}={v/g:.1f}" for name, v in sorted(c.items())), "(per GPU cycle)")
PY
