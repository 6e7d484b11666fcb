#!/bin/bash
# timing-only ablations of the three-wave kernel (libyue_hip_ablN.so, make EXTRA=-DYUE_ABL=N): with bit 1 (no version check)
# nothing waits for another wave, so an epoch's time is the pipeline's own throughput; further bits remove parts of waves L / S
set -o pipefail
O=gpurun_out/${1:-r4i}; mkdir -p $O
for n in 1 17 49; do
  echo "== ablation bits $n"
  YUE_LIB=yue_amd/csrc/libyue_hip_abl$n.so timeout -k 10 120 python tools/exact_probe.py c3 2 chain_split=1 chain_fast=1 2>&1 | tee $O/abl_c3_$n.log | tail -1
done
