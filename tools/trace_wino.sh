#!/bin/bash
mkdir -p implicit-depth_amd/_obj/abl
# Phase timeline of the Winograd conv kernel: tools/trace_wino.sh build (here) / run <cin> <cout> <H> <W> <tile_n> [B] (GPU box)
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DIDH_ABL_WINO_TRACE $EXTRA -c implicit-depth_amd/csrc/conv_wino.hip -o /tmp/conv_wino_trace.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls implicit-depth_amd/_obj/*.o | grep -v conv_wino.o) /tmp/conv_wino_trace.o -o implicit-depth_amd/_obj/abl/libidh_winotrace.so && echo built
else
  shift
  IDH_LIB=$PWD/implicit-depth_amd/_obj/abl/libidh_winotrace.so python tools/trace_wino.py "$@"
fi
