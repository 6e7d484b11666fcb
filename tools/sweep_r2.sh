#!/bin/bash
# round-2 sweep of the pipelined round kernel: events per wave (T) x round size (W) on C3
out=gpurun_out/$1; mkdir -p $out
shift
for cfg in "$@"; do
  t=${cfg%%:*}; w=${cfg##*:}
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline --tpw $t --round-events $w > $out/c3_t${t}_w${w}.json 2> $out/c3_t${t}_w${w}.err
  python - <<PY
import json
try:
    d=json.load(open('$out/c3_t${t}_w${w}.json'))
    print('T=$t W=$w  %.3e triplets/s  %.2f ms/epoch  launch %.1f us  frac %.3f' % (d['value'], d['ms_per_step'], 1e3*d['roofline']['avg_launch_ms'], d['roofline']['frac']))
except Exception as e:
    print('T=$t W=$w failed', e, open('$out/c3_t${t}_w${w}.err').read()[-500:])
PY
done
