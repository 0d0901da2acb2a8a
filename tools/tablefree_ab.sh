#!/bin/bash
# Round 5: bucket reduction of the table-free path -- segmented running sums per window (rounds 1-4: msm_reduce + msm_window_finish,
# BP_MSM_REDUCE=1 in the experiment build) against the bit-plane tree over the forest of W bucket sets + per-window Horner (shipped).
# Device / accumulate / tail ms of one MSM without tables, alternating, N rounds per size.    tools/tablefree_ab.sh [ROUNDS] [LOG_N ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-2}; shift
SIZES=${@:-"16 18 20 22"}
export BABY_PLONK_LIBRARY=exp
for lg in $SIZES; do
  for i in $(seq 1 $N); do
    for RED in 1 0; do
      echo "2^$lg  $([ $RED = 1 ] && echo 'running sums (r04)' || echo 'bit-plane tree    ')  $(BP_MSM_REDUCE=$RED python3 $R/tools/sweep_window_bits.py --log-n $lg --widths 1 --reps 6 2>&1 | grep device_ms | tail -1 | grep -o '"device_ms.*other_ms": [0-9.]*')"
    done
  done
done
