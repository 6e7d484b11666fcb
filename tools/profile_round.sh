#!/bin/bash
# usage: tools/profile_round.sh TAG   -- the evidence set of a round on one MI355X (run through gpurun):
# full GPU test suite, the default bench line, rocprofv3 kernel stats of the same command, the two PMC traffic passes,
# the traffic JSON derived from them, the driver's form of the bench command, the deviation-vs-W tables,
# the secondary workloads (random-factor C5, C2, C4 shard through the communicator path, FISM rounds), PMC passes of the scoring kernels.
tag=$1; out=gpurun_out/$tag; mkdir -p $out gpurun_out/pm
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests -q -m gpu > $out/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $out/gpu_tests.log; tail -2 $out/gpu_tests.log
python bench.py > $out/bench_c3.json 2> $out/bench_c3.err && echo bench ok
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/kt.log 2>&1 && cp $(find $out/kt -name '*kernel_stats.csv' | head -1) $out/bench_c3_kernel_stats.csv && rm -rf $out/kt && echo kernel stats ok
BENCH_ARGS="--steps 1 --warmup 1 --no-secondary" tools/pmc_pass.sh ${tag}_fetch FETCH_SIZE && BENCH_ARGS="--steps 1 --warmup 1 --no-secondary" tools/pmc_pass.sh ${tag}_write WRITE_SIZE TCC_EA0_ATOMIC_sum
W=$(python -c "import json; print(json.load(open('$out/bench_c3.json'))['config']['round_events'])") && python tools/traffic_json.py gpurun_out/pm/${tag}_fetch.txt gpurun_out/pm/${tag}_write.txt c3 $W "$tag, rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum, separate passes" > $out/traffic.json && echo traffic ok
python bench.py --steps 20 --warmup 5 > $out/bench_c3_steps20_warmup5.json 2> $out/bench_c3_steps20.err && echo driver-form bench ok
python tools/deviation_table.py c3 8192 57344 172032 344064 516096 > $out/deviation_c3.jsonl 2> $out/deviation_c3.err && python tools/deviation_table.py c2 8192 57344 114688 172032 229376 344064 > $out/deviation_c2.jsonl 2> $out/deviation_c2.err && echo deviation tables ok
python bench.py --workload c5 --steps 2 --warmup 1 > $out/bench_c5.json 2> $out/bench_c5.err && echo c5 ok
python bench.py --workload c2 --no-secondary > $out/bench_c2.json 2> $out/bench_c2.err && echo c2 ok
python bench.py --workload c4shard --force-comm --steps 2 --warmup 1 --no-cpu-baseline > $out/bench_c4shard.json 2> $out/bench_c4shard.err && echo c4shard ok
python bench.py --workload fism --steps 3 --warmup 1 > $out/bench_fism.json 2> $out/bench_fism.err && echo fism ok
BENCH_ARGS="--steps 2 --warmup 4" tools/pmc_pass.sh ${tag}_scan_mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS && BENCH_ARGS="--steps 2 --warmup 4" tools/pmc_pass.sh ${tag}_scan_gui GRBM_GUI_ACTIVE
