#!/bin/bash
# tools/build_variant.sh <tag> <extra hipcc -D flags...>  ->  ultrazoom_amd/libmewzoom_hip_<tag>.so
# e.g.  tools/build_variant.sh noW -DQ_ABLATE_W     (timing-only: conv3q_kernel without its weight DMA)
set -euo pipefail
tag=$1; shift
here="$(cd "$(dirname "$0")/../ultrazoom_amd/csrc" && pwd)"
mkdir -p "$here/build"
/opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -c "$here/mz_host.cpp" -o "$here/build/mz_host.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c "$here/mz_kernels.hip" -o "$here/build/mz_kernels_$tag.o" &
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c "$here/mz_conv3q.hip" -o "$here/build/mz_conv3q_$tag.o" &
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c "$here/mz_conv3r.hip" -o "$here/build/mz_conv3r_$tag.o" &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "$here/build/mz_kernels_$tag.o" "$here/build/mz_conv3q_$tag.o" "$here/build/mz_conv3r_$tag.o" "$here/build/mz_host.o" -o "$here/../libmewzoom_hip_$tag.so"
echo "built $tag"
