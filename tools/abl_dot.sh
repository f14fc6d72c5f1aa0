#!/bin/bash
mkdir -p implicit-depth_amd/_obj/abl
# ablation builds of the LDS-window dot-volume kernel: tools/abl_dot.sh build (here) / run [B] (GPU box)
cd "$(dirname "$0")/.."
VARS="${VARS:-NOSTAGE NOCOMPUTE NOBARRIER NOSTAGE,NOBARRIER}"
if [ "$1" = build ]; then
  for v in $VARS; do
    flags=""; for f in ${v//,/ }; do flags="$flags -DIDH_ABL_$f"; done
    name=${v//,/_}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $flags -c implicit-depth_amd/csrc/cost_volume_dot.hip -o /tmp/cvd_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls implicit-depth_amd/_obj/*.o | grep -v cost_volume_dot.o) /tmp/cvd_$name.o -o implicit-depth_amd/_obj/abl/libidh_abldot_$name.so && echo built $name
  done
else
  B=${2:-32}
  echo "== base"; python tools/perf_dot.py 3 $B
  for v in $VARS; do name=${v//,/_}; echo "== $name (timing only: results are meaningless)"; IDH_LIB=$PWD/implicit-depth_amd/_obj/abl/libidh_abldot_$name.so python tools/perf_dot.py 3 $B; done
fi
