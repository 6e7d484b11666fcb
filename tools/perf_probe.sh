#!/bin/bash
# usage: tools/perf_probe.sh "ENV=.. ENV2=.." ...   (each argument = one bench run with that environment)
for cfg in "$@"; do
  env $cfg timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/probe.log 2>&1
  python - "$cfg" <<'PY'
import json,sys
try:
    d=json.loads(open("gpurun_out/probe.log").read().strip().splitlines()[-1]); r=d["roofline"]
    print("%-40s value %.3e  ms/step %6.1f  K_A avg %6.1f us" % (sys.argv[1], d["value"], d["ms_per_step"], r["avg_launch_ms"]*1e3))
except Exception as e:
    print(sys.argv[1], "FAILED", e, open("gpurun_out/probe.log").read()[-300:])
PY
done
