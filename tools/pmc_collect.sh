#!/bin/bash
# Collects rocprofv3 evidence for one small forward (3 images = one micro-batch of the cfg3 workload) — run on the GPU box.
# usage: tools/pmc_collect.sh <outdir-under-gpurun_out> [workload]   (separate passes: counters never share a run with other traces)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-pmc}
WL=${2:-cfg3_1080p}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-microbench --no-secondary --images-per-gpu 3 --workload $WL"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-microbench --no-secondary --workload $WL > $OUT/stats.log 2>&1
echo "stats rc=$?"
run() { # name counters...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- $CMD > $OUT/$name.log 2>&1
  echo "$name rc=$?"
}
run sqA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
run sqB SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM
run grbm GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
