#!/bin/bash
# msm_accumulate cut into G workgroup generations (BP_MSM_CHUNK = 104 / G entries per lane at 2^20 points, grid G x),
# with and without the accumulation on a lowest-priority stream and everything else on highest-priority streams (BP_ACC_LOW_PRIORITY=1).
# (a) three 2^20 commitments back to back (one by one / lanes / batched), (b) bp_prove at 2^20 gates.  Experiment build, one box, alternating.
#   tools/generations_ab.sh [ROUNDS]
R=${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-2}
export BABY_PLONK_LIBRARY=exp
for i in $(seq 1 $N); do
  for PRIO in 0 1; do
    for CH in 0 52 26 13; do
      echo "== round $i  BP_ACC_LOW_PRIORITY=$PRIO  BP_MSM_CHUNK=$CH (G = $([ $CH = 0 ] && echo 1 || echo $((104 / CH))))"
      if [ $CH = 0 ]; then
        BP_ACC_LOW_PRIORITY=$PRIO python3 $R/tools/commit_batch_ab.py --log-n 20 --reps 8 2>&1 | grep commitments
        BP_ACC_LOW_PRIORITY=$PRIO python3 $R/tools/run_prove.py --log-n 20 --reps 4 2>&1 | tail -2
      else
        BP_ACC_LOW_PRIORITY=$PRIO BP_MSM_CHUNK=$CH python3 $R/tools/commit_batch_ab.py --log-n 20 --reps 8 2>&1 | grep commitments
        BP_ACC_LOW_PRIORITY=$PRIO BP_MSM_CHUNK=$CH python3 $R/tools/run_prove.py --log-n 20 --reps 4 2>&1 | tail -2
      fi
    done
  done
done
