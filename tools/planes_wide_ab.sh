#!/bin/bash
# A/B of the bit-plane tree's wide/narrow threshold (additions from which a level runs one addition per lane through HBM): device ms per MSM.
#   tools/planes_wide_ab.sh LOG_N WIDTH MIN [MIN ...]        (experiment build: BP_MSM_PLANES_WIDE_MIN)
export BABY_PLONK_LIBRARY=exp
R=${GRAFT_REPO_ROOT:-$(pwd)}
LG=$1; W=$2; shift; shift
for m in "$@"; do
  echo "wide_min $m: $(BP_MSM_PLANES_WIDE_MIN=$m python3 $R/tools/sweep_window_bits.py --log-n $LG --widths $W --reps 6 2>&1 | grep device_ms | tail -1)"
done
