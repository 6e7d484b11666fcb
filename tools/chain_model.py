#!/usr/bin/env python3
"""What bounds an exact epoch?  oracle/bpr_oracle.c: orc_dataflow_model on a BASELINE stream (CPU only):
python tools/chain_model.py c3|c2|c4shard  -- makespan for a grid of (workers, step, hop) in microseconds."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yue_amd import synth                      # noqa: E402
from oracle import loader                      # noqa: E402

W = {'c3': (1000000, 200000, 50, 128), 'c2': (100000, 50000, 50, 64), 'tiny': (20000, 5000, 20, 128), 'c4shard': (10000000, 125000, 6, 128)}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'c3'
    m, n, d, k = W[name]
    data = synth.make_arrays(m, n, d, seed=20260001)
    orc = loader.Oracle()
    ev_u = np.repeat(np.arange(m, dtype=np.int32), np.diff(data['ev_ptr']))
    j = orc.sample_counter(20260003, 0, ev_u, n, data['indptr'], data['indices'])
    depth, row_max = orc.dependency_depth(ev_u, data['ev_i'], j, m, n)
    print('%s: %d triplets, dependency depth %d, hottest row %d touches' % (name, len(ev_u), depth, row_max))
    grid = [eval(a) for a in sys.argv[2:]] or [(1024, 6.0, 0.54, 1.0), (1024, 6.0, 0.44, 1.0), (256, 6.0, 0.31, 1.3), (256, 6.0, 0.18, 1.3), (512, 6.0, 0.18, 1.3),
                                             (1024, 6.0, 0.18, 1.3), (256, 2.0, 0.18, 1.3), (256, 6.0, 0.18, 0.4), (1024, 6.0, 0.18, 0.4), (100000, 0.0, 0.18, 0.0)]
    for (workers, startup, step, hop) in grid:
        t, hops = orc.dataflow_model(ev_u, data['ev_i'], j, n, workers, startup, step, hop, hop)
        print('workers %6d  startup %.1f us  step %.2f us  hop %.2f us  ->  %.1f ms  (%.2e triplets/s), %d cross-run hops on the longest path' % (
            workers, startup, step, hop, t / 1e3, len(ev_u) / (t * 1e-6), hops), flush=True)


if __name__ == '__main__':
    main()
