#!/bin/bash
# secondary bench lines recorded under profiles/: tools/bench_matrix.sh <outdir>
OUT=gpurun_out/$1; mkdir -p $OUT
B="python bench.py --no-cpu-baseline --no-split-line"
$B --workload warp_match_dot --steps 30 | tail -1 > $OUT/warp_match_dot_k8_gb32.json
$B --workload warp_match_dot --views 7 --steps 30 | tail -1 > $OUT/warp_match_dot_k7_gb32.json
for m in fp32 f16x3; do
  $B --math $m --volume dot --steps 20 | tail -1 > $OUT/hot_path_dot_k8_gb32_$m.json
  $B --math $m --volume mlp --views 8 --steps 20 | tail -1 > $OUT/hot_path_mlp_k8_gb32_$m.json
  $B --math $m --planes 96 --steps 20 | tail -1 > $OUT/hot_path_mlp_k7_d96_gb32_$m.json
  for b in 1 4 8 16 32; do $B --math $m --batch $b --steps 20 | tail -1 > $OUT/hot_path_mlp_k7_gb${b}_$m.json; done
done
for f in $OUT/*.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read()); r=d["roofline"]
print(f"{sys.argv[1].split('/')[-1]:42s} fps {d['value']:9.1f}  ms/step {d['ms_per_step']:8.3f}  {r['kernel'][:28]:28s} {r['achieved']:8.1f} {r['unit']}  frac {r['frac']:.3f}")
PY
done
