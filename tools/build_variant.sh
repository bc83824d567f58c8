#!/bin/bash
# tools/build_variant.sh <tag> <extra hipcc -D flags...>  ->  ultrazoom_amd/libmewzoom_hip_<tag>.so
# e.g.  tools/build_variant.sh diag -DMZ_DIAG     (in-kernel cycle stamps of conv3r_kernel / conv3t_kernel: mz_diag.h, tools/stamp_probe_*.py)
set -euo pipefail
tag=$1; shift
here="$(cd "$(dirname "$0")/../ultrazoom_amd/csrc" && pwd)"
mkdir -p "$here/build"
/opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -c "$here/mz_host.cpp" -o "$here/build/mz_host.o"
units=(mz_kernels mz_conv3r mz_conv3t mz_probe)
pids=()
for u in "${units[@]}"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c "$here/$u.hip" -o "$here/build/${u}_$tag.o" &
    pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
objs=("$here/build/mz_host.o")
for u in "${units[@]}"; do objs+=("$here/build/${u}_$tag.o"); done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$here/../libmewzoom_hip_$tag.so"
echo "built $tag"
