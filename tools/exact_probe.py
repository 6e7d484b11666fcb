#!/usr/bin/env python3
"""Timing probe of the exact path (chain_kernels.hpp): python tools/exact_probe.py c3|c2|tiny [epochs] [NAME=VALUE ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yue_amd import synth                      # noqa: E402
from yue_amd import _shim                      # noqa: E402
from yue_amd._shim import Device               # noqa: E402
if os.environ.get('YUE_LIB'):
    _shim.LIB_PATH = os.environ['YUE_LIB']

W = {'c3': (1000000, 200000, 50, 128), 'c2': (100000, 50000, 50, 64), 'tiny': (20000, 5000, 20, 128), 'c4shard': (10000000, 125000, 6, 128)}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'tiny'
    epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    m, n, d, k = W[name]
    data = synth.make_arrays(m, n, d, seed=20260001)
    P0, Q0 = synth.init_factors(m, n, k, 20260002)
    dev = Device(0, raise_errors=True)
    dev.set_factors(P0, Q0)
    dev.set_interactions(data['indptr'], data['indices'], data['ev_ptr'], data['ev_i'])
    for kv in sys.argv[3:]:
        a, b = kv.split('=')
        dev.set_option(a, int(b))
    E = int(data['ev_ptr'][-1])
    dev.set_option('epoch_exact', 1)
    for ep in range(epochs):
        t0 = time.perf_counter()
        nll, sp, sq = dev.bpr_epoch(20260003, ep, 0, 0.02, 0.01, 0.01)
        dt = time.perf_counter() - t0
        print('%s exact epoch %d: %.1f ms  %.3e triplets/s  nll/triplet %.5f  (runs %d, waves %d; the dataflow launch alone %.1f ms)' % (
            name, ep, 1e3 * dt, E / dt, nll / E, dev.get_option('chain_last_runs'), dev.get_option('chain_last_waves'), 1e-3 * dev.get_option('chain_last_us')), flush=True)
    dev.set_option('epoch_exact', 0)
    if '--replay' in os.environ.get('PROBE', ''):
        j = dev.sample_negatives(20260003, 0)
        ev_u = np.repeat(np.arange(m, dtype=np.int32), np.diff(data['ev_ptr']))
        dev.set_factors(P0, Q0)
        t0 = time.perf_counter()
        dev.bpr_replay(ev_u, data['ev_i'], j, 0.02, 0.01, 0.01)
        dt = time.perf_counter() - t0
        print('%s replay of the same stream from host arrays: %.1f ms  %.3e triplets/s' % (name, 1e3 * dt, E / dt), flush=True)
    dev.close()


if __name__ == '__main__':
    main()
