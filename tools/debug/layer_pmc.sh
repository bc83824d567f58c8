#!/bin/bash
# GPU box: PMC counters of one layer under three builds/knobs
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
shape="$1"
for v in s16 s32 hack; do
  unset MZ_NO_S16 MEWZOOM_HIP_LIB
  [ $v = s32 ] && export MZ_NO_S16=1
  [ $v = hack ] && export MZ_NO_S16=1 MEWZOOM_HIP_LIB=$R/ultrazoom_amd/libmewzoom_hip_s16.so
  for pass in A B C; do
    case $pass in
      A) C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS";;
      B) C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VALU";;
      C) C="GRBM_GUI_ACTIVE";;
    esac
    rm -rf /tmp/lp_$v$pass
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/lp_$v$pass -- python3 $R/tools/debug/layer_time.py $shape 6 > /tmp/lp_$v$pass.log 2>&1
    f=$(find /tmp/lp_$v$pass -name "*counter_collection.csv" | head -1)
    echo "== $v $pass"
    python3 - "$f" <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'conv3' in r['Kernel_Name']:
        d[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in d.items():
    print('  %-32s %.4g (n=%d)' % (k, sum(v[1:]) / max(1, len(v) - 1), len(v)))
PY
  done
done
