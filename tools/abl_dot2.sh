#!/bin/bash
mkdir -p implicit-depth_amd/_obj/abl
# A/B builds of cv_dot_win_k with arbitrary -D flags: tools/abl_dot2.sh build "name:-DFLAG -DFLAG2" ... (here) / run [B] name ... (GPU box)
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  shift
  for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $flags -c implicit-depth_amd/csrc/cost_volume_dot.hip -o /tmp/cvd_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls implicit-depth_amd/_obj/*.o | grep -v cost_volume_dot.o) /tmp/cvd_$name.o -o implicit-depth_amd/_obj/abl/libidh_abldot_$name.so && echo built $name
  done
else
  B=${2:-32}; shift; shift
  echo "== HEAD"; python tools/perf_dot.py 3 $B
  for name in "$@"; do echo "== $name"; IDH_LIB=$PWD/implicit-depth_amd/_obj/abl/libidh_abldot_$name.so python tools/perf_dot.py 3 $B; done
fi
