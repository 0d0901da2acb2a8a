#!/bin/bash
# Round 5: order / parity of the digit widths of a three-pass transform.  Device ms of one transform, alternating rounds.
#   tools/ntt_order_ab.sh [ROUNDS]      (experiment build: BP_NTT_SPLIT)
export BABY_PLONK_LIBRARY=exp
N=${1:-3}
run() { python tools/run_msm.py --log-n 10 --reps 8 --ntt-log-n $1 2>&1 | grep "^ntt" | tail -4 | sed 's/.*device //' | tr '\n' ' '; echo; }
for round in $(seq 1 $N); do
while read lg splits; do
  for sp in $splits; do echo -n "2^$lg $sp: "; BP_NTT_SPLIT="$lg:$sp" run $lg; done
done <<LIST
20 7,6,7 6,6,8 6,8,6 5,7,8
21 7,7,7 7,6,8 6,7,8 8,6,7
22 8,7,7 8,6,8 6,8,8 7,7,8
LIST
done
