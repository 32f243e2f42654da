#!/bin/bash
# build_variant.sh <name> <source.hip> "<extra flags>": libmgnet_hip.so with ONE source compiled with extra flags -> ablib/libmgnet_hip_<name>.so
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; flags=$3
mkdir -p ablib/obj_$name
base=$(basename $src .hip)
CF="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Imgnet_amd/csrc -include mgnet_amd/csrc/mgn_launch.h"
/opt/rocm/bin/hipcc $CF $flags -c mgnet_amd/csrc/$base.hip -o ablib/obj_$name/$base.o
objs=""
for o in mgnet_amd/lib/obj/*.o; do
  b=$(basename $o)
  if [ "$b" == "$base.o" ]; then objs="$objs ablib/obj_$name/$base.o"; elif [ "$b" == "${base}_f16.o" ] && [ -f mgnet_amd/csrc/${base}_f16.hip ]; then
    /opt/rocm/bin/hipcc $CF $flags -c mgnet_amd/csrc/${base}_f16.hip -o ablib/obj_$name/${base}_f16.o; objs="$objs ablib/obj_$name/${base}_f16.o"
  else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o ablib/libmgnet_hip_$name.so
echo ablib/libmgnet_hip_$name.so
