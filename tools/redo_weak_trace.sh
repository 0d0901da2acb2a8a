#!/bin/bash
set -e
TAG=r05
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
P=$O/profiles_$TAG
mkdir -p $P
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_kt_weak2 -o b -- python3 $R/bench.py --steps 20 --warmup 5 --skip-cpu --prove-log-n 0 \
  --other-sizes --skip-seams --skip-pipelined --strong-log-n 0 > $P/${TAG}_bench_weak_under_rocprof.json 2> $O/${TAG}_kt_weak2.err
cd $R
cp "$(find $O/${TAG}_kt_weak2 -name b_kernel_stats.csv | head -1)" $P/${TAG}_bench_weak_kernel_stats.csv
python3 tools/kernel_stats_by_grid.py "$(find $O/${TAG}_kt_weak2 -name b_kernel_trace.csv | head -1)" > $P/${TAG}_bench_weak_kernel_stats_by_grid.csv
head -30 $P/${TAG}_bench_weak_kernel_stats_by_grid.csv
