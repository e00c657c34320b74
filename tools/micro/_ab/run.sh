#!/bin/bash
# A/B of two builds of the library in one gpurun call: the in-tree one and tools/micro/_ab/libswat_hip_old.so
R=${GRAFT_REPO_ROOT:-/root/repo}
cp $R/swat_amd/lib/libswat_hip.so /tmp/new.so
trap 'cp /tmp/new.so $R/swat_amd/lib/libswat_hip.so' EXIT   # an interrupt or a failing command must not leave the old build installed
for rep in 1 2; do
  for w in old new; do
    if [ $w = old ]; then cp $R/tools/micro/_ab/libswat_hip_old.so $R/swat_amd/lib/libswat_hip.so; else cp /tmp/new.so $R/swat_amd/lib/libswat_hip.so; fi
    echo "== $w"; "$@" 2>&1 | grep -v amdgpu.ids | tail -${TAILN:-3}
  done
done
cp /tmp/new.so $R/swat_amd/lib/libswat_hip.so
