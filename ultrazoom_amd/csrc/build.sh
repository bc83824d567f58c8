#!/bin/bash
# Builds libmewzoom_hip.so for gfx950 (MI355X). hipcc cross-compiles without a GPU.
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
out="$here/../libmewzoom_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
mkdir -p "$here/build"
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$here/mz_kernels.hip" -o "$here/build/mz_kernels.o" &
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$here/mz_conv3q.hip" -o "$here/build/mz_conv3q.o" &
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$here/mz_conv3r.hip" -o "$here/build/mz_conv3r.o" &
"$HIPCC" -O2 -std=c++17 -fPIC -c "$here/mz_host.cpp" -o "$here/build/mz_host.o" &
wait
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "$here/build/mz_kernels.o" "$here/build/mz_conv3q.o" "$here/build/mz_conv3r.o" "$here/build/mz_host.o" -o "$out"
echo "built $out"
# on-box MFMA peak micro-benchmark (bench.py's `roofline.measured_peak` leg)
mb="$here/../../tools/microbench"
"$HIPCC" --offload-arch=gfx950 -O3 "$mb/mb_mfma.hip" -o "$mb/mb_mfma"
echo "built $mb/mb_mfma"
