#!/bin/bash
# timing-only ablations of k_scan_filter on config 5 (factors of 25 epochs): one scan_probe.py run per diagnostic library
# (yue_amd/csrc: make EXTRA=-DYUE_FILTER_ABL=bits OBJDIR=build_fablN LIB=libyue_hip_fablN.so); lists are wrong by construction
out=gpurun_out/${1:-fabl}; mkdir -p $out
for n in 0 2 34 32 0 2 34; do
    lib=yue_amd/csrc/libyue_hip_fabl$n.so; [ $n = 0 ] && lib=yue_amd/csrc/libyue_hip.so
    [ -f $lib ] || continue
    YUE_LIB=$PWD/$lib PROBE_DEFAULT=1 timeout -k 10 300 python tools/scan_probe.py 25 > $out/abl$n.txt 2>&1 || { echo "ablation $n failed"; tail -3 $out/abl$n.txt; exit 1; }
    echo "ablation bits $n: $(tail -1 $out/abl$n.txt)"
done
