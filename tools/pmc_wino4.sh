#!/bin/bash
# PMC passes comparing the Winograd F(4x4) and F(2x2) kernels on one layer: tools/pmc_wino4.sh <tag> cin cout H W [B]
# (every pass under its own timeout: a counter name rocprofv3 does not know aborts the tool and leaves the child hanging)
export TMPDIR=/tmp
TAG=$1; CIN=$2; COUT=$3; H=$4; W=$5; B=${6:-32}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp
TMS=${TMS:-13 12}
for tm in $TMS; do
  ARGS="$CIN $COUT $H $W $tm 0 1 $B 6"
  run() { name=$1; shift; timeout 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/t${tm}_$name -o p -- python $GRAFT_REPO_ROOT/tools/one_conv.py $ARGS > $OUT/t${tm}_$name.log 2>&1; }
  run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE
  run sq2 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU
  run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR
  run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr
  run fetch FETCH_SIZE
  run write WRITE_SIZE
done
python - <<PY
import csv, glob, collections
for tm in [int(v) for v in "$TMS".split()]:
    print("== tile code", tm, "(13 = F(4x4), 12 = F(2x2))")
    for name in ["sq1","sq2","lds","tcp","fetch","write"]:
        fs = glob.glob("$OUT/t%d_%s/**/*counter_collection.csv" % (tm, name), recursive=True)
        if not fs: print(name, "no csv"); continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(fs[0])):
            if "conv3x3" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(f"{name:6s} {k:34s} n={len(v)} last={v[-1]:.5g}")
    fs = glob.glob("$OUT/t%d_sq1/**/*kernel_trace.csv" % tm, recursive=True)
    if fs:
        d = [ (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) for r in csv.DictReader(open(fs[0])) if "conv3x3" in r["Kernel_Name"]]
        print(f"kernel duration (profiled) last = {d[-1] / 1e3:.1f} us, n = {len(d)}")
PY
