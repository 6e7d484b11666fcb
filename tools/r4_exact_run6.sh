#!/bin/bash
set -o pipefail
O=gpurun_out/${1:-r4o}; mkdir -p $O
for w in c2 c4shard; do
for opts in "chain_split=0 chain_fast=0" "chain_split=0 chain_fast=1" "chain_split=1 chain_fast=0" "chain_split=1 chain_fast=1" "chain_split=1 chain_fast=1 chain_xcd=1"; do
  tag=$(echo $opts | tr ' =' '__')
  timeout -k 10 400 python tools/exact_probe.py $w 2 $opts > $O/probe_${w}_$tag.log 2>&1 || { tail -20 $O/probe_${w}_$tag.log; exit 1; }
  echo "== $w $opts"; tail -1 $O/probe_${w}_$tag.log
done; done
