#!/bin/bash
# usage: tools/sweep_opts.sh "ARGS1" "ARGS2" ...   one C3 bench line per argument string (extra bench.py arguments)
for cfg in "$@"; do
  timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $cfg > gpurun_out/sweep.log 2>&1
  python - "$cfg" <<'PY'
import json,sys
try:
    d=json.loads(open("gpurun_out/sweep.log").read().strip().splitlines()[-1]); r=d["roofline"]
    print("%-60s value %.3e  ms/step %6.1f  k_round avg %6.1f us  nll %.5f" % (sys.argv[1], d["value"], d["ms_per_step"], r["avg_launch_ms"]*1e3, d["config"]["final_nll_per_triplet"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e, open("gpurun_out/sweep.log").read()[-300:])
PY
done
