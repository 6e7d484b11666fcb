#!/bin/bash
# usage (on the GPU box, through gpurun): tools/ab.sh OUTDIR "tag1:bench args" "tag2:bench args" ...
# A/B runs of the C3 bench on ONE box (boxes differ by a few per cent): prints tag, ms/epoch, average round launch (us), roofline fraction, nll.
out=$1; shift; mkdir -p $out
for spec in "$@"; do
    tag=${spec%%:*}; args=${spec#*:}
    timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary $args > $out/bench_$tag.json 2> $out/bench_$tag.err || { echo "$tag FAILED"; tail -3 $out/bench_$tag.err; continue; }
    python - "$out/bench_$tag.json" "$tag" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d['roofline']
print('%-22s %8.3f ms/epoch  %7.2f us/launch  frac %.4f  nll %.10f' % (sys.argv[2], d['ms_per_step'], r['avg_launch_ms'] * 1e3, r['frac'], d['config']['final_nll_per_triplet']))
PY
done
