#!/bin/bash
export BABY_PLONK_LIBRARY=exp      # BP_* knobs are read by the experiment build only (make -C baby_plonk_rust_amd/csrc exp)
# A/B of the pass split of one transform size (BP_NTT_SPLIT="k:l1,l2[,l3]", read by csrc/ntt.hip); device ms, two readings each
run() { python tools/run_msm.py --log-n 10 --reps 5 --ntt-log-n $1 2>&1 | grep "^ntt" | tail -2 | sed 's/.*device //' | tr '\n' ' '; echo; }
while read lg splits; do
  echo -n "2^$lg default: "; run $lg
  for sp in $splits; do echo -n "2^$lg $sp: "; BP_NTT_SPLIT="$lg:$sp" run $lg; done
done <<LIST
16 6,5,5 8,8,0 9,7,0
18 6,6,6 7,6,5 9,9,0
19 7,6,6 10,9,0 8,6,5
20 7,7,6 8,6,6 8,8,4 9,6,5 6,7,7 10,5,5 7,6,7
21 7,7,7 8,7,6 9,6,6
22 8,7,7 10,6,6 9,7,6 7,7,8
24 8,8,8 10,7,7 9,8,7 10,10,4 9,9,6
LIST
