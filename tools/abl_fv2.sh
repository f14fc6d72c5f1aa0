#!/bin/bash
mkdir -p implicit-depth_amd/_obj/abl
# A/B builds of fv_mlp_k with arbitrary -D flags: tools/abl_fv2.sh build "name:-DFLAG ..." ... (here) / run name ... (GPU box)
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  shift
  for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $flags -c implicit-depth_amd/csrc/feature_volume.hip -o /tmp/fv_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls implicit-depth_amd/_obj/*.o | grep -v feature_volume.o) /tmp/fv_$name.o -o implicit-depth_amd/_obj/abl/libidh_ablfv_$name.so && echo built $name
  done
else
  shift
  echo "== HEAD"; python tools/perf_fv.py 32 7 64 10 | tail -1
  for name in "$@"; do echo "== $name"; IDH_LIB=$PWD/implicit-depth_amd/_obj/abl/libidh_ablfv_$name.so python tools/perf_fv.py 32 7 64 10 | tail -1; done
fi
