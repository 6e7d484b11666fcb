#!/bin/bash
# usage: BENCH_ARGS="--workload c5small --steps 1 --warmup 1" tools/pmc_pass.sh TAG COUNTER [COUNTER ...]
# one rocprofv3 --pmc pass over a short bench run (default: C3 training); summary -> gpurun_out/pm/TAG.txt
tag=$1; shift
mkdir -p gpurun_out/pm
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pm/$tag -- python3 bench.py ${BENCH_ARGS:---steps 1 --warmup 1} --no-cpu-baseline > gpurun_out/pm/$tag.log 2>&1 || { echo "pass $tag failed"; tail -5 gpurun_out/pm/$tag.log; exit 1; }
python tools/pmc_summary.py gpurun_out/pm/$tag > gpurun_out/pm/$tag.txt
rm -rf gpurun_out/pm/$tag
echo "pass $tag done"
