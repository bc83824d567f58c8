#!/bin/bash
# PMC comparison of two library builds: tools/pmc_compare_libs.sh <outdir> <lib1> <lib2> ...
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename $lib .so)
  export MEWZOOM_HIP_LIB=$R/ultrazoom_amd/$lib
  CMD="python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-microbench --images-per-gpu 3"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/$name/sqA -- $CMD > $OUT/$name.sqA.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/$name/grbm -- $CMD > $OUT/$name.grbm.log 2>&1
  echo "$name done"
done
