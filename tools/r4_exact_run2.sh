#!/bin/bash
# round-4 exact-path session 2: step / hop latencies of the chain kernels, ring depth, groups per CU
set -o pipefail
O=gpurun_out/${1:-r4b}; mkdir -p $O
for opts in "chain_split=0 chain_fast=0" "chain_split=0 chain_fast=1" "chain_split=1 chain_fast=0" "chain_split=1 chain_fast=1" "chain_split=1 chain_fast=1 chain_ring=16"; do
  tag=$(echo $opts | tr ' =' '__')
  timeout -k 10 300 python tools/chain_step_probe.py $opts > $O/step_$tag.log 2>&1 || { tail -20 $O/step_$tag.log; exit 1; }
  echo "== $opts"; cat $O/step_$tag.log
done
for opts in "chain_split=1 chain_fast=1 chain_ring=16" "chain_split=1 chain_fast=1 chain_waves=2" "chain_split=1 chain_fast=1 chain_waves=4" "chain_split=1 chain_fast=1 chain_waves=2 chain_ring=16"; do
  tag=$(echo $opts | tr ' =' '__')
  timeout -k 10 200 python tools/exact_probe.py c3 2 $opts > $O/probe_c3_$tag.log 2>&1 || { tail -20 $O/probe_c3_$tag.log; exit 1; }
  echo "== $opts"; tail -1 $O/probe_c3_$tag.log
done
