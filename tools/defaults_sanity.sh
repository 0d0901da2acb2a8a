#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
echo "== window widths at 2^20 and 2^21 (shipped library)"
python3 $R/tools/sweep_window_bits.py --log-n 20 --widths 18 19 20 21 22 2>&1 | grep device_ms
python3 $R/tools/sweep_window_bits.py --log-n 21 --widths 19 20 21 22 2>&1 | grep device_ms
echo "== concurrent provers per GPU"
for S in 1 2 3 4; do python3 $R/bench.py --prove-streams $S --prove-reps 4 --skip-cpu --skip-seams --other-sizes --strong-log-n 0 --steps 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('provers', $S, 'proofs/s', round(d['prove']['value'],2), 'latency ms', round(d['prove']['latency_ms_per_proof_single_prover'],2))"; done
echo "== NTT pass splits (experiment build)"
cd $R && bash tools/ntt_split_sweep.sh 2>&1 | grep -E "^2\^(20|22|24)"
