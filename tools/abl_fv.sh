#!/bin/bash
mkdir -p implicit-depth_amd/_obj/abl
# ablation builds of the f16x3 feature-volume kernel: tools/abl_fv.sh build (here) / run (GPU box)
cd "$(dirname "$0")/.."
VARS="${VARS:-NOMFMA NOGATHER NOMFMA,NOGATHER}"
if [ "$1" = build ]; then
  for v in $VARS; do
    flags=""; for f in ${v//,/ }; do flags="$flags -DIDH_ABL_$f"; done
    name=${v//,/_}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $flags -c implicit-depth_amd/csrc/feature_volume.hip -o /tmp/fv_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls implicit-depth_amd/_obj/*.o | grep -v feature_volume.o) /tmp/fv_$name.o -o implicit-depth_amd/_obj/abl/libidh_ablfv_$name.so && echo built $name
  done
else
  run() { python - <<'PY'
import torch, time, sys, os
sys.path.insert(0, os.getcwd())
import implicit_depth_amd.synthetic as syn
from implicit_depth_amd import cost_volume as cv
B, K, H, W, D = 32, 7, 96, 128, 64
dev = torch.device("cuda:0")
d = {k: v.to(dev) for k, v in syn.cost_volume_inputs(B, K, 16, H, W, seed=0).items()}
for math in ("fp32", "f16x3"):
    m = cv.FeatureVolumeManager(H, W, D, num_source_views=K).to(dev); syn.fill_state_dict(m.mlp, 99, gain=1.4); m.mlp_math = math
    for _ in range(2): m(**d)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): m(**d)
    e1.record(); torch.cuda.synchronize()
    print(f"  {math}: {e0.elapsed_time(e1)/5:.2f} ms per 32-frame volume (incl. layout + argmax)")
PY
  }
  echo "== base"; run
  for v in $VARS; do name=${v//,/_}; echo "== $name"; IDH_LIB=$PWD/implicit-depth_amd/_obj/abl/libidh_ablfv_$name.so run; done
fi
