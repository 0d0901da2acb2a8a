#!/bin/bash
# Same-box A/B of one experiment knob at one MSM size: tools/knob_ab.sh LOG_N WIDTH ROUNDS KNOB VALUE [VALUE ...]   (experiment build)
export BABY_PLONK_LIBRARY=exp
R=${GRAFT_REPO_ROOT:-$(pwd)}
LG=$1; W=$2; N=$3; K=$4; shift; shift; shift; shift
for i in $(seq 1 $N); do
  for v in "$@"; do
    echo "2^$LG $K=$v: $(env $K=$v python3 $R/tools/sweep_window_bits.py --log-n $LG --widths $W --reps 4 2>&1 | grep device_ms | tail -1 | grep -o '"device_ms.*other_ms": [0-9.]*')"
  done
done
