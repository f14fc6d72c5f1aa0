#!/bin/bash
# Round-end evidence run on the GPU box: tools/final_profiles.sh <tag>   (writes gpurun_out/<tag>/...)
export TMPDIR=/tmp
TAG=${1:-r02}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT/matrix
cd $ROOT
B="timeout 300 python bench.py --no-cpu-baseline --no-parity"
echo "== default bench line"; timeout 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "== rocprofv3 kernel stats: hot path"; ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_hot -o hot -- python $ROOT/bench.py --no-cpu-baseline --no-parity --no-extras --steps 30 --warmup 3 > $OUT/prof_hot.log 2>&1 )
echo "== rocprofv3 kernel stats: warp_match_dot"; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_dot -o dot -- python $ROOT/bench.py --workload warp_match_dot --no-cpu-baseline --steps 50 --warmup 5 > $OUT/prof_dot.log 2>&1 )
echo "== rocprofv3 kernel stats: temporal"; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_temporal -o t -- python $ROOT/bench.py --workload temporal --no-cpu-baseline --steps 48 --warmup 8 > $OUT/prof_temporal.log 2>&1 )
echo "== PMC HBM traffic"; tools/pmc_bench.sh ${TAG}_hot --no-parity --no-extras > $OUT/pmc_hot.txt 2>&1; tools/pmc_bench.sh ${TAG}_dot --workload warp_match_dot > $OUT/pmc_dot_traffic.txt 2>&1
echo "== PMC MFMA util"; tools/pmc_mfma.sh ${TAG} --no-parity --no-extras > $OUT/pmc_mfma.txt 2>&1
echo "== PMC dot kernels"; tools/pmc_dot.sh ${TAG}_win 3 32 > $OUT/pmc_dot_win.txt 2>&1; tools/pmc_dot.sh ${TAG}_quad 2 32 > $OUT/pmc_dot_quad.txt 2>&1
echo "== matrix"
$B --no-extras --workload warp_match_dot --steps 50 | tail -1 > $OUT/matrix/warp_match_dot_k8_gb32.json
$B --no-extras --workload warp_match_dot --views 7 --steps 50 | tail -1 > $OUT/matrix/warp_match_dot_k7_gb32.json
$B --no-extras --workload temporal --steps 64 | tail -1 > $OUT/matrix/temporal_k7_d96_b1.json
$B --no-extras --volume dot --steps 30 | tail -1 > $OUT/matrix/hot_path_dot_k8_gb32_fp32.json
$B --no-extras --planes 96 --steps 30 | tail -1 > $OUT/matrix/hot_path_mlp_k7_d96_gb32_fp32.json
$B --no-extras --no-head --steps 30 | tail -1 > $OUT/matrix/hot_path_mlp_k7_gb32_fp32_nohead.json
for b in 1 4 8 16; do $B --no-extras --batch $b --steps 30 | tail -1 > $OUT/matrix/hot_path_mlp_k7_gb${b}_fp32.json; done
for b in 1 4 32; do $B --no-extras --math f16x3 --batch $b --steps 30 | tail -1 > $OUT/matrix/hot_path_mlp_k7_gb${b}_f16x3.json; done
for f in $OUT/matrix/*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read()); r=d["roofline"]
    print(f"{sys.argv[1].split('/')[-1]:44s} fps {d['value']:9.1f}  ms/step {d['ms_per_step']:8.3f}  {r['kernel'][:30]:30s} {r['achieved']:8.1f} {r['unit']}  frac {r['frac']:.3f}")
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
done
for t in hot dot temporal; do f=$(find $OUT/prof_$t -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { echo "-- $t"; head -12 "$f" | cut -c1-170; }; done
cat $OUT/pmc_mfma.txt | tail -12; tail -8 $OUT/pmc_hot.txt; tail -5 $OUT/pmc_dot_traffic.txt
