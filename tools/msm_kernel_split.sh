#!/bin/bash
# per-kernel durations of one MSM size / table width: tools/msm_kernel_split.sh LOG_N WIDTH [WIDTH ...]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
LG=$1; shift
cd /tmp
for w in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/split_${LG}_$w -o t -- python3 $R/tools/sweep_window_bits.py --log-n $LG --widths $w --reps 4 > $R/gpurun_out/split_${LG}_$w.log 2>&1
  grep device_ms $R/gpurun_out/split_${LG}_$w.log
  python3 $R/tools/kernel_stats_by_grid.py $R/gpurun_out/split_${LG}_$w/t_kernel_trace.csv | grep -v "srs_\|fr_synth" | head -22
done
