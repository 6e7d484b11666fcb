#!/bin/bash
# timing-only ablations of the batch round kernel (results are wrong by construction; never used by the package)
out=gpurun_out/$1; mkdir -p $out
for dbg in 0 1 5 3 7; do
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --opt round_kernel=1 --opt debug_skip=$dbg > $out/abl_$dbg.json 2> $out/abl_$dbg.err
  python - <<PY
import json
try:
    d=json.load(open('$out/abl_$dbg.json'))
    print('debug_skip=$dbg  %.3e triplets/s  %.2f ms/epoch  launch %.1f us' % (d['value'], d['ms_per_step'], 1e3*d['roofline']['avg_launch_ms']))
except Exception as e:
    print('debug_skip=$dbg failed', e, open('$out/abl_$dbg.err').read()[-300:])
PY
done
