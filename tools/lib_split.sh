#!/bin/bash
# per-kernel durations of one MSM size / table width for a given build of the library: tools/lib_split.sh LIB.so LOG_N WIDTH [WIDTH ...]
# (width 1 = no tables).  Output under gpurun_out/split_<tag>_<log_n>_<width>.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
LIB=$1; LG=$2; shift; shift
TAG=$(basename $LIB .so)
export BABY_PLONK_LIBRARY=$R/$LIB
cd /tmp
for w in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/split_${TAG}_${LG}_$w -o t -- python3 $R/tools/sweep_window_bits.py --log-n $LG --widths $w --reps 6 > $R/gpurun_out/split_${TAG}_${LG}_$w.log 2>&1
  echo "== $TAG 2^$LG width $w"; grep device_ms $R/gpurun_out/split_${TAG}_${LG}_$w.log
  python3 $R/tools/kernel_stats_by_grid.py $R/gpurun_out/split_${TAG}_${LG}_$w/t_kernel_trace.csv | grep -v "srs_\|fr_synth" | head -24
done
