#!/bin/bash
# GPU box: time the baseline and each ablation variant on 3 images of the cfg3 workload, per-launch CSVs.
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/abl
for m in base "$@"; do
  lib=$R/ultrazoom_amd/libmewzoom_hip.so
  [ "$m" != base ] && lib=$R/ultrazoom_amd/libmewzoom_hip_ab$m.so
  MEWZOOM_HIP_LIB=$lib timeout -k 10 200 python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --images-per-gpu 3 \
     --dump-launches $R/gpurun_out/abl/$m.csv > $R/gpurun_out/abl/$m.log 2>&1
  echo "$m rc=$? $(tail -1 $R/gpurun_out/abl/$m.log | grep -o '"achieved": [0-9.]*')"
done
