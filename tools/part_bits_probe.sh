#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
export BABY_PLONK_LIBRARY=exp
for lg in 16 18 20 22; do
  for RB in default 9 10 11 12; do
    if [ $RB = default ]; then unset BP_MSM_RADIX_BITS; else export BP_MSM_RADIX_BITS=$RB; fi
    echo "2^$lg no tables, partition bits $RB: $(python3 $R/tools/sweep_window_bits.py --log-n $lg --widths 1 --reps 6 2>&1 | grep device_ms | tail -1 | grep -o '"device_ms.*other_ms": [0-9.]*')"
  done
done
