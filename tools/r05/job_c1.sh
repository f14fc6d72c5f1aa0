#!/bin/bash
# batched 1x1 projection chunks in the direct LDS kernel: old library vs new, plan time at 1 / 4 / 32 frames + bit check of the outputs
export TMPDIR=/tmp
OLD=$PWD/implicit-depth_amd/lib/ab/libidh_old.so
mkdir -p gpurun_out/c1
for b in 1 4 32; do
  for r in 1 2; do
    for l in old new; do
      if [ $l = old ]; then export IDH_LIB=$OLD; else unset IDH_LIB; fi
      IDH_LEVELS_ORDER=plan timeout 280 python tools/perf_levels.py $b > gpurun_out/c1/levels_b${b}_${l}.txt 2>&1
      echo "$l $(grep units= gpurun_out/c1/levels_b${b}_${l}.txt)"
    done
  done
done
for l in old new; do
  if [ $l = old ]; then export IDH_LIB=$OLD; else unset IDH_LIB; fi
  timeout 200 python - <<'PY'
import torch, hashlib, sys
sys.path.insert(0, '.')
import implicit_depth_amd.synthetic as syn
from implicit_depth_amd import networks as net
dec = net.BDDecoderPP([24, 64, 128, 256, 384]); syn.fill_state_dict(dec, seed=21); dec.cuda()
for B in (1, 3):
    feats = [t.cuda() for t in syn.encoder_pyramid(B, 384, 512, seed=5, channels=(24, 64, 128, 256, 384))]
    out = dec(feats)
    print(B, {k: hashlib.md5(v.cpu().numpy().tobytes()).hexdigest()[:10] for k, v in sorted(out.items())})
PY
done
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -3
