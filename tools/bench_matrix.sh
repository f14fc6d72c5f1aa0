#!/bin/bash
# secondary bench lines recorded under profiles/: tools/bench_matrix.sh <outdir>
OUT=gpurun_out/$1; mkdir -p $OUT
python bench.py --no-cpu-baseline --workload warp_match_dot --steps 30 | tail -1 > $OUT/warp_match_dot_k8_gb32.json
python bench.py --no-cpu-baseline --workload warp_match_dot --views 7 --steps 30 | tail -1 > $OUT/warp_match_dot_k7_gb32.json
python bench.py --no-cpu-baseline --volume dot --steps 20 | tail -1 > $OUT/hot_path_dot_k8_gb32.json
python bench.py --no-cpu-baseline --volume mlp --views 8 --steps 20 | tail -1 > $OUT/hot_path_mlp_k8_gb32.json
python bench.py --no-cpu-baseline --planes 96 --steps 20 | tail -1 > $OUT/hot_path_mlp_k7_d96_gb32.json
for b in 1 4 8; do python bench.py --no-cpu-baseline --batch $b --steps 20 | tail -1 > $OUT/hot_path_mlp_k7_gb$b.json; done
for f in $OUT/*.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read()); r=d["roofline"]
print(f"{sys.argv[1].split('/')[-1]:38s} fps {d['value']:9.1f}  ms/step {d['ms_per_step']:8.3f}  {r['kernel'][:22]:22s} {r['achieved']:8.1f} {r['unit']}  frac {r['frac']:.3f}")
PY
done
