#!/bin/bash
# GPU box: per-kernel time of one layer under three builds/knobs (rocprofv3 kernel trace)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
shape="$1"
for v in s16 s32 hack; do
  unset MZ_NO_S16 MEWZOOM_HIP_LIB
  [ $v = s32 ] && export MZ_NO_S16=1
  [ $v = hack ] && export MZ_NO_S16=1 MEWZOOM_HIP_LIB=$R/ultrazoom_amd/libmewzoom_hip_s16.so
  rm -rf /tmp/lt_$v
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lt_$v -- python3 $R/tools/debug/layer_time.py $shape 30 > /tmp/lt_$v.log 2>&1
  f=$(find /tmp/lt_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v $shape ($f)"; [ -n "$f" ] && grep -E "conv3" "$f" | sed -E "s/mz::ConvArgs//" | cut -c1-160 || tail -5 /tmp/lt_$v.log
done
