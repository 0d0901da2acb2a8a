#!/bin/bash
# per-kernel durations of one skewed distribution: tools/skew_split.sh KIND [WINDOW_BITS]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
K=$1; W=${2:-0}
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/skew_${K}_$W -o t -- python3 $R/tools/skew_timing.py --only $K --window-bits $W > $R/gpurun_out/skew_${K}_$W.log 2>&1
grep "tables=True" $R/gpurun_out/skew_${K}_$W.log
python3 $R/tools/kernel_stats_by_grid.py $R/gpurun_out/skew_${K}_$W/t_kernel_trace.csv | grep -v "srs_\|fr_synth" | head -24
