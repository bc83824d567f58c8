#!/bin/bash
# GPU box: one layer's kernel time with and without an environment knob:  layer_env.sh "<B H W cin cout>" KNOB
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
shape="$1"; knob="$2"
for v in off on off on; do
  unset $knob
  [ $v = on ] && export $knob=1
  rm -rf /tmp/le_$v
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/le_$v -- python3 $R/tools/debug/layer_time.py $shape 30 > /tmp/le_$v.log 2>&1
  f=$(find /tmp/le_$v -name "*kernel_stats.csv" | head -1)
  echo "== $knob=$v $shape"; grep -E "conv3" "$f" | sed -E "s/mz::ConvArgs//" | cut -c1-100
done
