#!/bin/bash
export BABY_PLONK_LIBRARY=exp      # BP_* knobs are read by the experiment build only (make -C baby_plonk_rust_amd/csrc exp)
# A/B of the NTT tile width (columns per LDS tile) and of the pass split; knobs are environment variables read by csrc/ntt.hip
run() { python tools/run_msm.py --log-n 10 --reps 5 --ntt-log-n $1 2>&1 | grep "^ntt" | tail -2 | tr '\n' ' '; echo; }
for lg in 16 18 19 20 21 22 24; do echo -n "default            2^$lg: "; run $lg; done
for lg in 19 20; do for c in 3 2; do echo -n "three passes, l<=7 cols_log=$c 2^$lg: "; BP_NTT_THREE_PASS_FROM=19 BP_NTT_COLS_LOG_L7=$c run $lg; done; done
for lg in 21 22; do for c in 3 2; do echo -n "l<=7 cols_log=$c     2^$lg: "; BP_NTT_COLS_LOG_L7=$c run $lg; done; done
echo -n "l=8 cols_log=3 (old) 2^24: "; BP_NTT_COLS_LOG_L8=3 run 24
echo -n "l=9 cols_log=2 (old) 2^18: "; BP_NTT_COLS_LOG_L9=2 run 18
