#!/bin/bash
# GPU box: the round-4 evidence in one go (run from the repository root; outputs under gpurun_out/r04/, copied into profiles/ by hand).
#   part A: driver-shaped bench lines + per-launch tables;  part B: rocprofv3 kernel stats + PMC passes;  part C: in-kernel stamps
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04
mkdir -p $OUT
part=${1:-A}
if [ "$part" = A ]; then
  python3 $R/bench.py --steps 10 --warmup 3 --dump-launches $OUT/per_launch_cfg3.csv > $OUT/bench_cfg3_1080p.json 2> $OUT/bench_cfg3.err; echo "bench cfg3 rc=$?"
  python3 $R/bench.py --workload cfg2 --steps 10 --warmup 3 --no-cpu-baseline --no-microbench --dump-launches $OUT/per_launch_cfg2.csv > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err; echo "bench cfg2 rc=$?"
elif [ "$part" = B ]; then
  bash $R/tools/pmc_collect.sh r04/pmc cfg3_1080p
  python3 $R/tools/pmc_traffic.py $OUT/pmc > $OUT/pmc_traffic_conv3x3.json; echo "traffic rc=$?"
  python3 $R/tools/pmc_summary.py $OUT/pmc 14 > $OUT/pmc_summary_per_kernel.txt; echo "summary rc=$?"
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg2 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-microbench --no-secondary --workload cfg2 > $OUT/stats_cfg2.log 2>&1; echo "stats cfg2 rc=$?"
else
  (MZ_DEBUG_STAMPS=1 MEWZOOM_HIP_LIB=$R/ultrazoom_amd/libmewzoom_hip_diag.so timeout -k 10 300 python3 $R/tools/stamp_probe_t.py 2>&1 | grep -v amdgpu.ids) > $OUT/stamp_probe_conv3t.txt; echo "stamps t rc=$?"
  (MZ_DEBUG_STAMPS=1 MEWZOOM_HIP_LIB=$R/ultrazoom_amd/libmewzoom_hip_diag.so STAMP_CASES="[(3, 1080, 1920, 96, 192, 1), (3, 1080, 1920, 192, 96, 2), (3, 540, 960, 192, 384, 1), (3, 135, 240, 1536, 768, 0), (8, 67, 120, 384, 768, 1)]" timeout -k 10 300 python3 $R/tools/stamp_probe_r.py 2>&1 | grep -v amdgpu.ids) > $OUT/stamp_probe_conv3r.txt; echo "stamps r rc=$?"
fi
