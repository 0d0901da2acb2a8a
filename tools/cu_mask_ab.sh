#!/bin/bash
# Round 5: msm_accumulate on a stream with a CU mask that leaves one CU in every (KEEP + 1) to the other streams (BP_ACC_CU_KEEP, experiment build):
# three 2^20 commitments (one by one / lanes / batched) and bp_prove at 2^20 gates.   tools/cu_mask_ab.sh [ROUNDS]
R=${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-2}
export BABY_PLONK_LIBRARY=exp
for i in $(seq 1 $N); do
  for KEEP in 0 31 15 7 3; do
    echo "== round $i  BP_ACC_CU_KEEP=$KEEP ($([ $KEEP = 0 ] && echo 'no mask' || echo "accumulate on $KEEP of every $((KEEP + 1)) CUs"))"
    if [ $KEEP = 0 ]; then
      python3 $R/tools/commit_batch_ab.py --log-n 20 --reps 8 2>&1 | grep commitments
      python3 $R/tools/run_prove.py --log-n 20 --reps 4 2>&1 | tail -2
    else
      BP_ACC_CU_KEEP=$KEEP python3 $R/tools/commit_batch_ab.py --log-n 20 --reps 8 2>&1 | grep commitments
      BP_ACC_CU_KEEP=$KEEP python3 $R/tools/run_prove.py --log-n 20 --reps 4 2>&1 | tail -2
    fi
  done
done
