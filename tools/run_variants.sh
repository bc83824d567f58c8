#!/bin/bash
# GPU box: per-launch timing of library variants (tags) on 3 images of the cfg3 workload
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/var
for m in "$@"; do
  lib=$R/ultrazoom_amd/libmewzoom_hip_$m.so
  [ "$m" = base ] && lib=$R/ultrazoom_amd/libmewzoom_hip.so
  MEWZOOM_HIP_LIB=$lib timeout -k 10 200 python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --images-per-gpu 3 \
     --dump-launches $R/gpurun_out/var/$m.csv > $R/gpurun_out/var/$m.log 2>&1
  echo "$m rc=$? $(tail -1 $R/gpurun_out/var/$m.log | grep -o '"achieved": [0-9.]*')"
done
