#!/bin/bash
# tools/micro/_ab/variant.sh <name> <file.hip> [-DMACRO=..]...: an alternate build of ONE translation unit linked with the in-tree
# objects -> tools/micro/_ab/libs/lib_<name>.so (git-ignored; travels to the GPU box, which gpurun_out/ does not; select it with SWAT_HIP_LIB=$PWD/tools/micro/_ab/libs/lib_<name>.so)
set -e
R=$(cd "$(dirname "$0")/../../.." && pwd)
name=$1; src=$2; shift 2
mkdir -p $R/tools/micro/_ab/libs /tmp/_variant_$name
o=/tmp/_variant_$name/${src%.hip}.o
(cd $R/swat_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fvisibility=hidden -Wno-unused-function "$@" -c $src -o $o)
objs=$(ls $R/swat_amd/csrc/_obj/*.o | grep -v "/${src%.hip}.o" | grep -v "/${REPLACES:-__none__}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/micro/_ab/libs/lib_$name.so $objs $o
echo "built tools/micro/_ab/libs/lib_$name.so"
