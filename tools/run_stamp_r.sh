#!/bin/bash
# GPU box: stamp probe of conv3r_kernel for library variants (tags)
R=${GRAFT_REPO_ROOT:-/root/repo}
for m in "$@"; do
  echo "######## $m"
  MZ_DEBUG_STAMPS=1 MEWZOOM_HIP_LIB=$R/ultrazoom_amd/libmewzoom_hip_$m.so timeout -k 10 200 python3 $R/tools/stamp_probe_r.py 2>&1 | grep -v amdgpu.ids
done
