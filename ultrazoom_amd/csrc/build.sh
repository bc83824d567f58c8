#!/bin/bash
# Builds libmewzoom_hip.so for gfx950 (MI355X). hipcc cross-compiles without a GPU.
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
out="$here/../libmewzoom_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
mkdir -p "$here/build"
# one compile per translation unit, in parallel; every PID is waited for on its own so that the FIRST broken unit fails the build
# with its own compiler output (a bare `wait` returns 0 whatever the children did, and the failure surfaced only at link time)
units=(mz_kernels mz_conv3r mz_conv3t mz_probe)
pids=()
for u in "${units[@]}"; do
    "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$here/$u.hip" -o "$here/build/$u.o" &
    pids+=($!)
done
"$HIPCC" -O2 -std=c++17 -fPIC -c "$here/mz_host.cpp" -o "$here/build/mz_host.o" &
pids+=($!)
names=("${units[@]}" mz_host)
fail=0
for i in "${!pids[@]}"; do
    if ! wait "${pids[$i]}"; then
        echo "build.sh: compiling ${names[$i]} FAILED" >&2
        fail=1
    fi
done
[ "$fail" -eq 0 ] || exit 1
objs=()
for u in "${names[@]}"; do objs+=("$here/build/$u.o"); done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$out"
echo "built $out"
# on-box MFMA peak micro-benchmark (bench.py's `roofline.measured_peak` leg)
mb="$here/../../tools/microbench"
"$HIPCC" --offload-arch=gfx950 -O3 "$mb/mb_mfma.hip" -o "$mb/mb_mfma"
echo "built $mb/mb_mfma"
