#!/bin/bash
# GPU box: tools/ab_libs.sh tagA tagB [bench.py arguments, default: --images-per-gpu 3]  -- per-layer A/B of two library builds (tag "base" = the shipped libmewzoom_hip.so), two
# alternating rounds of `bench.py --images-per-gpu 3 --dump-launches`, compared with tools/cmp_launches.py
R=${GRAFT_REPO_ROOT:-/root/repo}
A=$1; B=$2; shift 2
EXTRA="${*:---images-per-gpu 3}"
mkdir -p $R/gpurun_out/ab
for i in 1 2; do for m in $A $B; do
  lib=$R/ultrazoom_amd/libmewzoom_hip_$m.so
  [ "$m" = base ] && lib=$R/ultrazoom_amd/libmewzoom_hip.so
  MEWZOOM_HIP_LIB=$lib timeout -k 10 200 python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-microbench $EXTRA \
     --dump-launches $R/gpurun_out/ab/$m$i.csv > $R/gpurun_out/ab/$m$i.log 2>&1 || exit 1
  echo "$m$i $(tail -1 $R/gpurun_out/ab/$m$i.log | grep -o '"ms_per_step": [0-9.]*')"
done; done
python3 $R/tools/cmp_launches.py $R/gpurun_out/ab/${A}1.csv $R/gpurun_out/ab/${B}1.csv
python3 $R/tools/cmp_launches.py $R/gpurun_out/ab/${A}2.csv $R/gpurun_out/ab/${B}2.csv
