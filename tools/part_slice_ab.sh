#!/bin/bash
# A/B of the partition sort's slice size (scalars per workgroup of msm_part_scatter) at one MSM size: per-kernel durations under rocprofv3.
#   tools/part_slice_ab.sh LOG_N WIDTH SLICE [SLICE ...]        (experiment build: the knob BP_MSM_PART_SLICE is read by it only)
export TMPDIR=/tmp BABY_PLONK_LIBRARY=exp
R=${GRAFT_REPO_ROOT:-$(pwd)}
LG=$1; W=$2; shift; shift
cd /tmp
for s in "$@"; do
  export BP_MSM_PART_SLICE=$s
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/slice_${LG}_$s -o t -- python3 $R/tools/sweep_window_bits.py --log-n $LG --widths $W --reps 4 > $R/gpurun_out/slice_${LG}_$s.log 2>&1
  echo "== slice $s: $(grep device_ms $R/gpurun_out/slice_${LG}_$s.log | tail -1)"
  python3 $R/tools/kernel_stats_by_grid.py $R/gpurun_out/slice_${LG}_$s/t_kernel_trace.csv | grep "msm_part\|msm_radix_final"
done
