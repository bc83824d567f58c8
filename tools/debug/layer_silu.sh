#!/bin/bash
# GPU box: cost of the SiLU epilogue on one layer (rocprofv3 kernel trace)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
shape="$1"
for v in 1 0; do
  rm -rf /tmp/ls_$v
  MZ_LAYER_SILU=$v timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ls_$v -- python3 $R/tools/debug/layer_time.py $shape 30 > /tmp/ls_$v.log 2>&1
  f=$(find /tmp/ls_$v -name "*kernel_stats.csv" | head -1)
  echo "== silu=$v $shape"; grep -E "conv3" "$f" | sed -E "s/mz::ConvArgs//" | cut -c1-110
done
