export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
timeout -k 10 300 python3 $R/tools/sweep_window_bits.py --log-n 20 --widths 16 273 274 275 276
for w in 274 275; do
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/naf_$w -o t -- python3 $R/tools/sweep_window_bits.py --log-n 20 --widths $w --reps 3 > /dev/null 2>&1
python3 $R/tools/kernel_stats_by_grid.py $R/gpurun_out/naf_$w/t_kernel_trace.csv | grep -v "srs_\|fr_synth" | head -16
done
