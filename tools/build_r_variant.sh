#!/bin/bash
# tools/build_r_variant.sh <tag> <extra hipcc -D flags...>  ->  ultrazoom_amd/libmewzoom_hip_<tag>.so
# Like build_variant.sh, but only mz_conv3r.hip is rebuilt with the flags (the other objects come from csrc/build.sh).
set -euo pipefail
tag=$1; shift
here="$(cd "$(dirname "$0")/../ultrazoom_amd/csrc" && pwd)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c "$here/mz_conv3r.hip" -o "$here/build/mz_conv3r_$tag.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "$here/build/mz_kernels.o" "$here/build/mz_conv3q.o" "$here/build/mz_conv3r_$tag.o" "$here/build/mz_host.o" -o "$here/../libmewzoom_hip_$tag.so"
echo "built $tag"
