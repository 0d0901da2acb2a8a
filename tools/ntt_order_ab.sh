#!/bin/bash
# Round 5: order of the digit widths of a three-pass transform (default: larger first).  Device ms of one transform, alternating, three rounds.
export BABY_PLONK_LIBRARY=exp
run() { python tools/run_msm.py --log-n 10 --reps 8 --ntt-log-n $1 2>&1 | grep "^ntt" | tail -4 | sed 's/.*device //' | tr '\n' ' '; echo; }
for round in 1 2 3; do
while read lg splits; do
  for sp in $splits; do echo -n "2^$lg $sp: "; BP_NTT_SPLIT="$lg:$sp" run $lg; done
done <<LIST
20 7,7,6 7,6,7 6,7,7
22 8,7,7 7,8,7 7,7,8
23 8,8,7 8,7,8 7,8,8
LIST
done
