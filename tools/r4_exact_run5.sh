#!/bin/bash
set -o pipefail
O=gpurun_out/${1:-r4f}; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_exact.py -x -q -m gpu -s > $O/exact_tests.log 2>&1 || { tail -30 $O/exact_tests.log; exit 1; }
tail -3 $O/exact_tests.log
for opts in "chain_split=1 chain_fast=0" "chain_split=1 chain_fast=1" "chain_split=1 chain_fast=0 chain_xcd=1" "chain_split=1 chain_fast=1 chain_xcd=1"; do
  tag=$(echo $opts | tr ' =' '__')
  timeout -k 10 300 python tools/chain_step_probe.py $opts > $O/step_$tag.log 2>&1 || { tail -20 $O/step_$tag.log; exit 1; }
  echo "== $opts"; cat $O/step_$tag.log
  timeout -k 10 200 python tools/exact_probe.py c3 2 $opts > $O/probe_c3_$tag.log 2>&1 || { tail -20 $O/probe_c3_$tag.log; exit 1; }
  tail -1 $O/probe_c3_$tag.log
  YUE_LIB=yue_amd/csrc/libyue_hip_chainstats.so timeout -k 10 200 python tools/exact_probe.py c3 1 $opts > $O/stats_c3_$tag.log 2>&1 || { tail -20 $O/stats_c3_$tag.log; exit 1; }
  grep "stats\]" $O/stats_c3_$tag.log
done
