#!/bin/bash
# usage: tools/profile_round.sh TAG [STAGE ...]   -- the evidence set of a round on one MI355X (run through gpurun; a gpurun call is
# limited to 20 minutes, so the set comes in stages, default all):
#   tests      full GPU test suite + smoke()
#   bench      the default bench line and the driver's form of the command (--steps 20 --warmup 5)
#   kernels    rocprofv3 kernel stats of the bench command, the two PMC traffic passes, the traffic JSON derived from them
#   deviation  distance of the S-round semantics from the sequential loop as a function of W (C3, C2)
#   secondary  random-factor C5, C2, C4 shard through the communicator path, c3wide, FISM rounds
#   scan       PMC passes of the scoring kernels (secondary.c5 of the default bench)
#   exact      rocprofv3 kernel stats of an exact epoch (secondary.exact of the default bench runs it)
tag=$1; shift; stages="${*:-tests bench kernels deviation secondary scan exact}"
out=gpurun_out/$tag; mkdir -p $out gpurun_out/pm
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for st in $stages; do case $st in
tests)
    python -m pytest tests -q -m gpu > $out/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $out/gpu_tests.log; tail -2 $out/gpu_tests.log
    python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -1 $out/smoke.log ;;
bench)
    python bench.py > $out/bench_c3.json 2> $out/bench_c3.err && echo bench ok
    python bench.py --steps 20 --warmup 5 > $out/bench_c3_steps20_warmup5.json 2> $out/bench_c3_steps20.err && echo driver-form bench ok ;;
kernels)
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/kt.log 2>&1 && cp $(find $out/kt -name '*kernel_stats.csv' | head -1) $out/bench_c3_kernel_stats.csv && rm -rf $out/kt && echo kernel stats ok
    BENCH_ARGS="--steps 1 --warmup 1 --no-secondary" tools/pmc_pass.sh ${tag}_fetch FETCH_SIZE && BENCH_ARGS="--steps 1 --warmup 1 --no-secondary" tools/pmc_pass.sh ${tag}_write WRITE_SIZE TCC_EA0_ATOMIC_sum
    W=$(python -c "from yue_amd._shim import Device; import numpy as np; d = Device(0, raise_errors=True); d.set_factors(np.zeros((8, 128), np.float32), np.zeros((200000, 128), np.float32)); print(d.default_round_events())") && python tools/traffic_json.py gpurun_out/pm/${tag}_fetch.txt gpurun_out/pm/${tag}_write.txt c3 $W "$tag, rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum, separate passes" > $out/traffic.json && echo traffic ok ;;
deviation)
    python tools/deviation_table.py c3 8192 57344 172032 344064 516096 > $out/deviation_c3.jsonl 2> $out/deviation_c3.err && python tools/deviation_table.py c2 8192 57344 114688 172032 229376 344064 > $out/deviation_c2.jsonl 2> $out/deviation_c2.err && echo deviation tables ok ;;
secondary)
    python bench.py --workload c5 --steps 2 --warmup 1 > $out/bench_c5.json 2> $out/bench_c5.err && echo c5 ok
    python bench.py --workload c2 --no-secondary > $out/bench_c2.json 2> $out/bench_c2.err && echo c2 ok
    python bench.py --workload c4shard --force-comm --steps 2 --warmup 1 --no-cpu-baseline > $out/bench_c4shard.json 2> $out/bench_c4shard.err && echo c4shard ok
    python bench.py --workload c4shard --steps 2 --warmup 1 --no-cpu-baseline > $out/bench_c4shard_nocomm.json 2> $out/bench_c4shard_nocomm.err && echo c4shard without communicator ok
    python bench.py --workload c3wide --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $out/bench_c3wide.json 2> $out/bench_c3wide.err && echo c3wide ok
    python bench.py --workload fism --steps 3 --warmup 1 > $out/bench_fism.json 2> $out/bench_fism.err && echo fism ok ;;
scan)
    # counters of the scoring kernels on all 1M users of config 5: random factors (every tile scored) and the factors 25 epochs leave
    for ep in 0 25; do
        PROBE_DEFAULT=1 timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d gpurun_out/pm/${tag}_scan_sq_$ep -- python3 tools/scan_probe.py $ep > gpurun_out/pm/${tag}_scan_sq_$ep.log 2>&1 && python tools/pmc_summary.py gpurun_out/pm/${tag}_scan_sq_$ep > gpurun_out/pm/${tag}_scan_sq_$ep.txt && rm -rf gpurun_out/pm/${tag}_scan_sq_$ep && echo "scan SQ pass ($ep epochs) done"
        PROBE_DEFAULT=1 timeout -k 10 400 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pm/${tag}_scan_gui_$ep -- python3 tools/scan_probe.py $ep > gpurun_out/pm/${tag}_scan_gui_$ep.log 2>&1 && python tools/pmc_summary.py gpurun_out/pm/${tag}_scan_gui_$ep > gpurun_out/pm/${tag}_scan_gui_$ep.txt && rm -rf gpurun_out/pm/${tag}_scan_gui_$ep && echo "scan GUI pass ($ep epochs) done"
        grep "epochs," gpurun_out/pm/${tag}_scan_gui_$ep.log
    done
    PROBE_DEFAULT=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks -- python3 tools/scan_probe.py 25 > $out/ks.log 2>&1 && cp $(find $out/ks -name '*kernel_stats.csv' | head -1) $out/scan_25epochs_kernel_stats.csv && rm -rf $out/ks && echo "scan kernel stats (25 epochs) done" ;;
exact)
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kx -- python3 tools/exact_probe.py c3 2 > $out/kx.log 2>&1 && cp $(find $out/kx -name '*kernel_stats.csv' | head -1) $out/exact_c3_kernel_stats.csv && rm -rf $out/kx && echo exact kernel stats ok ;;
*) echo "unknown stage $st" ;;
esac; done
