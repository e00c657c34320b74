#!/bin/bash
# Host-side AddressSanitizer + UBSan build of the extension (the device code is compiled as usual, uninstrumented: GPU ASan needs
# xnack+ and is not available on this pool).  -> swat_amd/lib/libswat_hip_asan.so (git-ignored), which
# tests/test_sanitizer_host.py drives through every host-only path it can reach without a GPU (argument checks, dispatch and
# tile-planning logic, workspace sizing, descriptor-cache bookkeeping up to the first HIP call).  Takes ~6 minutes on 8 cores.
set -e
cd "$(dirname "$0")/../swat_amd/csrc"
OBJ=_obj_asan
mkdir -p $OBJ ../lib
FLAGS="--offload-arch=gfx950 -O1 -g -fPIC -std=c++17 -fvisibility=hidden -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -fno-sanitize-recover=undefined"
n=0
for f in *.hip; do
  o=$OBJ/${f%.hip}.o
  if [ ! -f $o ] || [ $f -nt $o ] || [ -n "$(find . ../../include -maxdepth 1 \( -name '*.hpp' -o -name '*.h' \) -newer $o 2>/dev/null)" ]; then
    /opt/rocm/bin/hipcc $FLAGS -c $f -o $o &
    n=$((n + 1))
    if [ $((n % 6)) = 0 ]; then wait; fi
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -fno-gpu-sanitize -shared-libsan -o ../lib/libswat_hip_asan.so $OBJ/*.o
echo "built swat_amd/lib/libswat_hip_asan.so"
