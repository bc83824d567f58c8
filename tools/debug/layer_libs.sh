#!/bin/bash
# GPU box: one layer's kernel time under several library builds:  layer_libs.sh "<B H W cin cout>" tag1 tag2 ...  (base = shipped)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
shape="$1"; shift
for v in "$@" "$@"; do
  unset MEWZOOM_HIP_LIB
  [ $v != base ] && export MEWZOOM_HIP_LIB=$R/ultrazoom_amd/libmewzoom_hip_$v.so
  rm -rf /tmp/ll_$v
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ll_$v -- python3 $R/tools/debug/layer_time.py $shape 30 > /tmp/ll_$v.log 2>&1
  f=$(find /tmp/ll_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v $shape: $(grep -E "conv3" "$f" | sed -E 's/.*\)",//' | cut -d, -f1-3)"
done
