#!/bin/bash
set -o pipefail
O=gpurun_out/${1:-r4q}; mkdir -p $O
for opts in "round_user_seq=0" "round_user_seq=1 round_fast=1" "round_user_seq=1 round_fast=0"; do
  o=""; for kv in $opts; do o="$o --opt $kv"; done
  echo "== c3 $opts"
  timeout -k 10 300 python bench.py --workload c3 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary $o 2> $O/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3e triplets/s  %.2f ms/epoch  frac %.3f  nll/triplet %.6f' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['final_nll_per_triplet']))" || { tail -5 $O/err.log; exit 1; }
done
for opts in "round_user_seq=0" "round_user_seq=1 round_fast=1"; do
  o=""; for kv in $opts; do o="$o --opt $kv"; done
  echo "== c2 $opts"
  timeout -k 10 300 python bench.py --workload c2 --steps 10 --warmup 2 --no-cpu-baseline --no-secondary $o 2> $O/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3e triplets/s  %.2f ms/epoch  frac %.3f  nll/triplet %.6f' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['final_nll_per_triplet']))" || { tail -5 $O/err.log; exit 1; }
done
