#!/bin/bash
# Collects the round's profiles on the GPU box (run through gpurun from the repo root) and writes the JUDGED summaries straight into
# profiles/ under gpurun_out/profiles_$TAG/ (gpurun merges gpurun_out back; copy that directory's files into profiles/ and commit):
#   rocprofv3 kernel statistics of the headline legs, of the 2^24 legs and of a 2^20-gate proof; PMC passes (FETCH_SIZE and
#   WRITE_SIZE separately, counters only) for the MSM and NTT at 2^20 and 2^24 -> ${TAG}_hbm_traffic.json; the register-only
#   microbenchmarks (tools/ubench_g1add, tools/ubench_fr29) -> ${TAG}_ubench_valu_floor.txt; the bench line itself.
# Everything bench.py reads back (traffic, instruction mix, register-only rates) is produced here from ONE commit.
set -e
TAG=${1:-r05}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
P=$O/profiles_$TAG
mkdir -p $P
cd /tmp
# 0. register-only microbenchmarks (binaries built in the container: see the Build line at the top of each .hip)
{
  echo "# Register-only microbenchmarks of the two inner operations (no LDS, no memory): what the VALU alone allows on one MI355X (256 CUs, 2.4 GHz)."
  echo "# tools/ubench_g1add.hip  -- g1_add_mixed28 (the loop body of msm_accumulate), loop-carried accumulator and point; commit $(cat $R/.commit_for_profiles 2>/dev/null)"
  $R/tools/ubench_g1add | grep "g1_add_mixed28"
  echo "# tools/ubench_fr29.hip  -- fr29_butterfly (NTT)"
  $R/tools/ubench_fr29 | grep -i "butterfl"
} > $P/${TAG}_ubench_valu_floor.txt 2>&1
# 1. the weak-scaling legs alone (2^20 MSM with and without tables + 2^20 NTT): msm_accumulate's average here is the bench line's kernel_ms
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_kt_weak -o b -- python3 $R/bench.py --steps 20 --warmup 5 --skip-cpu --prove-log-n 0 \
  --other-sizes --skip-seams --skip-pipelined --strong-log-n 0 > $P/${TAG}_bench_weak_under_rocprof.json 2> $O/${TAG}_kt_weak.err
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
find_csv() { find $O/$1 -name "$2" | head -1; }
cp "$(find_csv ${TAG}_kt_weak b_kernel_stats.csv)" $P/${TAG}_bench_weak_kernel_stats.csv
python3 tools/kernel_stats_by_grid.py "$(find_csv ${TAG}_kt_weak b_kernel_trace.csv)" > $P/${TAG}_bench_weak_kernel_stats_by_grid.csv
cp "$(find_csv ${TAG}_kt_strong b_kernel_stats.csv)" $P/${TAG}_2p24_kernel_stats.csv
python3 tools/kernel_stats_by_grid.py "$(find_csv ${TAG}_kt_strong b_kernel_trace.csv)" > $P/${TAG}_2p24_kernel_stats_by_grid.csv
cp "$(find_csv ${TAG}_kt_prove p_kernel_stats.csv)" $P/${TAG}_prove_2p20_kernel_stats.csv
for lg in 20 24; do
  cp "$(find_csv ${TAG}_pmc_f$lg f_counter_collection.csv)" $P/${TAG}_pmc_f${lg}_counter_collection.csv
  cp "$(find_csv ${TAG}_pmc_w$lg w_counter_collection.csv)" $P/${TAG}_pmc_w${lg}_counter_collection.csv
done
python3 tools/pmc_traffic.py $P/${TAG}_pmc_f20_counter_collection.csv $P/${TAG}_pmc_w20_counter_collection.csv --tag 2p20 --window-bits 20 --out $P/${TAG}_hbm_traffic.json
python3 tools/pmc_traffic.py $P/${TAG}_pmc_f24_counter_collection.csv $P/${TAG}_pmc_w24_counter_collection.csv --tag 2p24 --window-bits 22 --out $P/${TAG}_hbm_traffic.json --merge
grep "msm 2^\|ntt 2^" $O/${TAG}_kt_strong.log | tail -3
# 5. the bench line of this commit, unprofiled, reading the files produced above once they are in profiles/
cp $P/${TAG}_hbm_traffic.json $P/${TAG}_ubench_valu_floor.txt $R/profiles/ 2>/dev/null || true
python3 bench.py > $P/${TAG}_bench_final.json 2> $O/${TAG}_bench_final.err
echo done
