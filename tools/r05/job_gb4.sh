#!/bin/bash
# kernel trace of the 4-frame batch: per-kernel totals and the idle time between kernels of one step
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/gb4; mkdir -p $OUT
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o gb4 -- python $ROOT/bench.py --no-cpu-baseline --no-split-line --no-extras --batch 4 --steps 30 --warmup 5 > $OUT/prof.log 2>&1
tail -1 $OUT/prof.log | cut -c1-200
cd $ROOT
python - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/gb4/prof/**/*kernel_trace.csv',recursive=True)[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r['Start_Timestamp']))
# last 30 steps: find step boundaries by the first kernel name of a step (pointwise_nchw_k)
idx=[i for i,r in enumerate(rows) if 'pointwise_nchw_k' in r['Kernel_Name']]
a,b=idx[-11],idx[-1]
seg=rows[a:b]
busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in seg)
span=int(rows[b]['Start_Timestamp'])-int(seg[0]['Start_Timestamp'])
print(f"10 steps: span {span/1e7:.3f} ms/step, kernel busy {busy/1e7:.3f} ms/step, launches/step {len(seg)/10:.0f}")
tot=collections.defaultdict(lambda:[0,0])
for r in seg:
    k=r['Kernel_Name'].replace('(anonymous namespace)::','')[:60]; t=tot[k]; t[0]+=1; t[1]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
for k,(n,t) in sorted(tot.items(),key=lambda kv:-kv[1][1])[:22]: print(f"{k:60s} {n/10:6.1f} x {t/n/1e3:8.1f} us = {t/1e7:6.3f} ms/step")
gaps=[int(seg[i+1]['Start_Timestamp'])-int(seg[i]['End_Timestamp']) for i in range(len(seg)-1)]
print("gap total/step %.3f ms; gaps>10us: %d/step"%(sum(g for g in gaps if g>0)/1e7, sum(1 for g in gaps if g>10000)/10))
PY
