#!/bin/bash
# Timing-only variants of the library (results are WRONG by construction): tools/build_ablations.sh 1 2 4 7
set -euo pipefail
here="$(cd "$(dirname "$0")/../ultrazoom_amd/csrc" && pwd)"
mkdir -p "$here/build"
[ -f "$here/build/mz_host.o" ] || /opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -c "$here/mz_host.cpp" -o "$here/build/mz_host.o"
for m in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DMZ_ABLATE=$m -c "$here/mz_kernels.hip" -o "$here/build/mz_kernels_ab$m.o" &&
    { [ -f "$here/build/mz_conv3q.o" ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$here/mz_conv3q.hip" -o "$here/build/mz_conv3q.o"; } &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "$here/build/mz_kernels_ab$m.o" "$here/build/mz_conv3q.o" "$here/build/mz_host.o" -o "$here/../libmewzoom_hip_ab$m.so" && echo "built ab$m" ) &
done
wait
