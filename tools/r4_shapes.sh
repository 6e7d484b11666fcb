#!/bin/bash
# the S-round epoch path on other shapes, with sequential user rows (k_round_u) and without (k_round_m)
set -o pipefail
O=gpurun_out/${1:-r4s2}; mkdir -p $O
for w in c4shard c3wide c2; do for seq in 1 0; do
  echo "== $w round_user_seq=$seq"
  timeout -k 10 400 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --opt round_user_seq=$seq 2> $O/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3e triplets/s  %.2f ms/epoch  frac %.3f  W %d  kernel %s' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['round_events'], d['roofline']['kernel'][:20]))" || { tail -5 $O/err.log; exit 1; }
done; done
