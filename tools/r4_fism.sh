#!/bin/bash
# round 4, FISM round form: parity tests, then the bench line with and without rows stored in place (and round 3's shuffle sum is gone in both)
tag=${1:-r04_f}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_fism.py -q -m gpu > $out/fism_tests.log 2>&1; echo "pytest rc=$?" >> $out/fism_tests.log; tail -3 $out/fism_tests.log
grep -q "rc=0" $out/fism_tests.log || exit 1
timeout -k 10 300 python bench.py --workload fism --steps 3 --warmup 1 > $out/bench_fism.json 2> $out/bench_fism.err && echo fism ok
timeout -k 10 300 python bench.py --workload fism --steps 3 --warmup 1 --no-cpu-baseline --opt fism_inplace=0 > $out/bench_fism_inplace0.json 2> $out/bench_fism_inplace0.err && echo fism inplace0 ok
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kf -- python3 bench.py --workload fism --steps 2 --warmup 1 --no-cpu-baseline > $out/kf.log 2>&1 && cp $(find $out/kf -name '*kernel_stats.csv' | head -1) $out/fism_kernel_stats.csv && rm -rf $out/kf && echo fism kernel stats ok
python - $tag <<'PY'
import json
for f in ('bench_fism', 'bench_fism_inplace0'):
    try:
        d = json.loads(open('gpurun_out/%s/%s.json' % (__import__("sys").argv[1], f)).read().strip().splitlines()[-1])
        print(f, '%.3e draws/s' % d['value'], '%.2f ms' % d['ms_per_step'], d['config']['final_half_sq_error'])
    except Exception as e:
        print(f, 'unreadable', e)
PY
head -8 $out/fism_kernel_stats.csv
