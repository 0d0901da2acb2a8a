#!/bin/bash
# Same-box A/B of several builds of the library: device / accumulate / tail ms of one MSM size, alternating, N rounds.
#   tools/lib_ab.sh LOG_N WIDTH ROUNDS LIB.so [LIB.so ...]        (paths relative to the repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}
LG=$1; W=$2; N=$3; shift; shift; shift
for i in $(seq 1 $N); do
  for lib in "$@"; do
    echo "$lib $(BABY_PLONK_LIBRARY=$R/$lib python3 $R/tools/sweep_window_bits.py --log-n $LG --widths $W --reps 6 2>&1 | grep device_ms | tail -1 | grep -o '"device_ms.*other_ms": [0-9.]*')"
  done
done
