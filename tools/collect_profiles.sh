#!/bin/bash
# Collects the round's profiles on the GPU box (run through gpurun from the repo root); results land under gpurun_out/$TAG_*,
# the summaries that are judged get copied into profiles/ by hand (see README).
#   rocprofv3 kernel statistics of the headline legs, of the 2^24 legs and of a 2^20-gate proof; PMC passes (FETCH_SIZE and
#   WRITE_SIZE separately, counters only) for the MSM and NTT at 2^20 and 2^24.
set -e
TAG=${1:-r04}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp
# 1. the weak-scaling legs alone (2^20 MSM with and without tables + 2^20 NTT): msm_accumulate's average here is the bench line's kernel_ms
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_kt_weak -o b -- python3 $R/bench.py --steps 20 --warmup 5 --skip-cpu --prove-log-n 0 \
  --other-sizes --skip-seams --strong-log-n 0 > $O/${TAG}_bench_weak_under_rocprof.json 2> $O/${TAG}_kt_weak.err
# 2. the 2^24 legs (strong scaling at N = 1: MSM with the auto table width + NTT)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_kt_strong -o b -- python3 $R/tools/run_msm.py --log-n 24 --reps 4 --tables 0 --ntt-log-n 24 \
  > $O/${TAG}_kt_strong.log 2>&1
# 3. one 2^20-gate proof
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_kt_prove -o p -- python3 $R/tools/run_prove.py --log-n 20 > $O/${TAG}_kt_prove.log 2>&1
# 4. PMC traffic
for lg in 20 24; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_f$lg -o f -- python3 $R/tools/run_msm.py --log-n $lg --reps 2 --tables 0 --ntt-log-n $lg > $O/${TAG}_pmc_f$lg.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_w$lg -o w -- python3 $R/tools/run_msm.py --log-n $lg --reps 2 --tables 0 --ntt-log-n $lg > $O/${TAG}_pmc_w$lg.log 2>&1
done
cd $R
grep "msm 2^\|ntt 2^" $O/${TAG}_kt_strong.log | tail -3
echo done
