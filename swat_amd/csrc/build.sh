#!/bin/bash
# Builds libswat_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
OUT=../lib
mkdir -p $OUT _obj
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -fvisibility=hidden -Wall -Wno-unused-function"
pids=()
for f in *.hip; do
  o=_obj/${f%.hip}.o
  stale=0
  [ -f $o ] || stale=1
  for dep in $f *.hpp ../../include/*.h build.sh; do [ $dep -nt $o ] && stale=1; done
  if [ $stale = 1 ]; then
    echo "hipcc $f"
    /opt/rocm/bin/hipcc $FLAGS -c $f -o $o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libswat_hip.so _obj/*.o
echo "built $OUT/libswat_hip.so"
