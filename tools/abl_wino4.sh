#!/bin/bash
# ablation builds of the Winograd F(4x4) kernel: tools/abl_wino4.sh build (here) / run (GPU box).  Results of these builds are
# meaningless; only their timing is read (profiles/r04/experiments.md).
cd "$(dirname "$0")/.."
mkdir -p implicit-depth_amd/_obj/abl
VARS="${VARS:-NOXFORM NOEPI NOA NOB NOHALO NOA,NOB,NOXFORM,NOHALO}"
if [ "$1" = build ]; then
  for v in $VARS; do
    name=${v//,/_}; defs=""; for d in ${v//,/ }; do defs="$defs -DIDH_ABL_W4_$d"; done
    F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wno-unused-function $defs -fno-slp-vectorize -mllvm -amdgpu-prealloc-sgpr-spill-vgprs"
    /opt/rocm/bin/hipcc $F -c implicit-depth_amd/csrc/conv_wino4.hip -o /tmp/w4_$name.o && /opt/rocm/bin/hipcc $F -c implicit-depth_amd/csrc/conv_wino4p.hip -o /tmp/w4p_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls implicit-depth_amd/_obj/*.o | grep -v "conv_wino4.o\|conv_wino4p.o") /tmp/w4_$name.o /tmp/w4p_$name.o -o implicit-depth_amd/_obj/abl/libidh_ablw4_$name.so && echo built $name
  done
elif [ "$1" = trace ]; then
  shift
  IDH_LIB=$PWD/implicit-depth_amd/_obj/abl/libidh_ablw4_TRACE.so python tools/trace_wino4.py "$@" 2>&1 | grep -v amdgpu.ids
else
  export LAYERS=${LAYERS:-0,2} VARIANTS=${VARIANTS:-wino4}
  echo "== shipped"; python tools/perf_wino4.py 32 2 2>&1 | grep -v amdgpu.ids
  for v in $VARS; do name=${v//,/_}; echo "== $name"; IDH_LIB=$PWD/implicit-depth_amd/_obj/abl/libidh_ablw4_$name.so python tools/perf_wino4.py 32 2 2>&1 | grep -v amdgpu.ids; done
fi
