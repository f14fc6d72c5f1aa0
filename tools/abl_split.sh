#!/bin/bash
mkdir -p implicit-depth_amd/_obj/abl
# ablation builds of the split-bf16 conv kernel: tools/abl_split.sh build   (here)  /  run (on the GPU box)
cd "$(dirname "$0")/.."
VARS="${VARS:-NORES,NOSTORE NOLOAD NOLOAD,NORES,NOSTORE}"
if [ "$1" = build ]; then
  for v in $VARS; do
    flags=""; for f in ${v//,/ }; do flags="$flags -DIDH_ABL_$f"; done
    name=${v//,/_}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $flags -c implicit-depth_amd/csrc/conv_split.hip -o /tmp/conv_split_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls implicit-depth_amd/_obj/*.o | grep -v conv_split.o) /tmp/conv_split_$name.o -o implicit-depth_amd/_obj/abl/libidh_abl_$name.so && echo built $name
  done
else
  echo "== base"; python tools/perf_split.py 32 | cut -c1-30,100-200
  for v in $VARS; do name=${v//,/_}; echo "== $name"; IDH_LIB=$PWD/implicit-depth_amd/_obj/abl/libidh_abl_$name.so python tools/perf_split.py 32 | cut -c1-30,100-200; done
fi
