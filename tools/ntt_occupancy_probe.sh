#!/bin/bash
export BABY_PLONK_LIBRARY=exp      # BP_* knobs are read by the experiment build only (make -C baby_plonk_rust_amd/csrc exp)
# How a pass of 2^7 x 8 tiles scales with tiles per CU: the same kernel on 256, 512, 1024, 2048 and 4096 tiles (transforms of
# 2^18 .. 2^22 split as 7 + 7 + rest), swizzled (four workgroups per CU) and padded (three); rocprofv3 kernel durations.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
for mode in 1 0; do
  for spec in "18:7,7,4" "19:7,7,5" "20:7,7,6" "21:7,7,7" "22:7,7,8"; do
    lg=${spec%%:*}
    d=$R/gpurun_out/nttocc_${mode}_$lg
    BP_NTT_SWIZZLE=$mode BP_NTT_SPLIT=$spec rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python3 $R/tools/run_msm.py --log-n 10 --reps 5 --ntt-log-n $lg > /dev/null 2>&1
    echo "swizzle=$mode 2^$lg ($spec):"
    grep "ntt_pass" $d/t_kernel_stats.csv | awk -F'","' '{gsub(/"/,"",$1); split($1,a,"("); printf "   %-28s calls %s avg_us %.1f min_us %.1f\n", a[1], $2, $4/1000, $6/1000}'
  done
done
