#!/bin/bash
mkdir -p implicit-depth_amd/_obj/abl
# ablation + trace builds of the Winograd conv kernel: tools/abl_wino.sh build (here) / run cin cout H W tile_n [B] (GPU box)
cd "$(dirname "$0")/.."
VARS="${VARS:-BASE NODMA NOXFORM NOSTORE NODMA,NOXFORM,NOSTORE}"
if [ "$1" = build ]; then
  for v in $VARS; do
    flags="-DIDH_ABL_WINO_TRACE"; for f in ${v//,/ }; do flags="$flags -DIDH_ABL_WINO_$f"; done
    name=${v//,/_}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $flags -c implicit-depth_amd/csrc/conv_wino.hip -o /tmp/cw_$name.o 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls implicit-depth_amd/_obj/*.o | grep -v conv_wino.o) /tmp/cw_$name.o -o implicit-depth_amd/_obj/abl/libidh_ablwino_$name.so && echo built $name
  done
else
  shift
  for v in $VARS; do name=${v//,/_}; echo "== $name"; IDH_LIB=$PWD/implicit-depth_amd/_obj/abl/libidh_ablwino_$name.so python tools/trace_wino.py "$@" 2>&1 | grep -v amdgpu.ids; done
fi
