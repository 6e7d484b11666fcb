#!/bin/bash
set -o pipefail
O=gpurun_out/${1:-r4h}; mkdir -p $O
for pool in 128 1024 4096; do for ring in 8 16; do
  PROBE_POOL=$pool PROBE_STEP_ONLY=1 timeout -k 10 120 python tools/chain_step_probe.py chain_split=1 chain_fast=1 chain_ring=$ring 2>&1 | tee -a $O/pool.log | tail -1
done; done
for pool in 128 4096; do
  PROBE_POOL=$pool PROBE_STEP_ONLY=1 timeout -k 10 120 python tools/chain_step_probe.py chain_split=1 chain_fast=1 chain_xcd=1 chain_waves=8 2>&1 | tee -a $O/pool.log | tail -1
  PROBE_POOL=$pool PROBE_STEP_ONLY=1 YUE_LIB=yue_amd/csrc/libyue_hip_abl3.so timeout -k 10 120 python tools/chain_step_probe.py chain_split=1 chain_fast=1 2>&1 | tee -a $O/pool.log | tail -1
done
