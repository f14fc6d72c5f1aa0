#!/bin/bash
# ablation builds of the fp32 feature-volume kernel fv_mlp_k: tools/abl_fv32.sh build (here) / run (GPU box: times each with tools/perf_fv.py).
# Results of these builds are meaningless; only their timing is read (profiles/r05/experiments.md).
cd "$(dirname "$0")/.."
mkdir -p implicit-depth_amd/_obj/abl
VARS="${VARS:-NOTAPS NOPRO NOBLEND NOACT NOL3 NORAY NOLDSA NOTAPS,NOPRO,NOBLEND,NOACT,NOL3,NORAY NOTAPS,NOPRO,NOBLEND,NOACT,NOL3,NORAY,NOLDSA}"
if [ "$1" = build ]; then
  for v in $VARS; do
    name=${v//,/_}; defs=""; for d in ${v//,/ }; do case $d in VG*) defs="$defs -DIDH_ABL_FV_VALU_PER_GROUP=${d#VG}";; *) defs="$defs -DIDH_ABL_FV2_$d";; esac; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wno-unused-function $defs -c implicit-depth_amd/csrc/feature_volume.hip -o /tmp/fv32_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls implicit-depth_amd/_obj/*.o | grep -v feature_volume.o) /tmp/fv32_$name.o -o implicit-depth_amd/_obj/abl/libidh_ablfv32_$name.so && echo built $name
  done
else
  echo "== shipped"; python tools/perf_fv.py 32 7 64 10 2>&1 | grep -v amdgpu.ids
  for v in $VARS; do name=${v//,/_}; echo "== $name"; IDH_LIB=$PWD/implicit-depth_amd/_obj/abl/libidh_ablfv32_$name.so python tools/perf_fv.py 32 7 64 10 2>&1 | grep -v amdgpu.ids; done
fi
