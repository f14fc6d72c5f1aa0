#!/bin/bash
mkdir -p implicit-depth_amd/_obj/abl
# A/B builds of conv_wino.hip with arbitrary -D flags: tools/abl_wino2.sh build "name:-DFLAG ..." ... (here) / run name ... (GPU box; LAYERS env selects perf_wino layers)
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  shift
  for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $flags -c implicit-depth_amd/csrc/conv_wino.hip -o /tmp/wino_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls implicit-depth_amd/_obj/*.o | grep -v conv_wino.o) /tmp/wino_$name.o -o implicit-depth_amd/_obj/abl/libidh_ablwino_$name.so && echo built $name
  done
else
  shift
  echo "== HEAD"; python tools/perf_wino.py 32 3 | grep -v amdgpu
  for name in "$@"; do echo "== $name"; IDH_LIB=$PWD/implicit-depth_amd/_obj/abl/libidh_ablwino_$name.so python tools/perf_wino.py 32 3 | grep -v amdgpu; done
fi
