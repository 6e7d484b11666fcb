#!/usr/bin/env python3
"""One exact epoch with the one-wave kernel (chain_split = 0) and the wave-group kernel (chain_split = 1) on the same problem: results must be bit-equal."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yue_amd import synth
from yue_amd._shim import Device
for (m, n, d, k) in [(3000, 2000, 20, 128), (5000, 64, 12, 64), (700, 900, 70, 10), (400, 300, 25, 200), (20000, 5000, 20, 128)]:
    data = synth.make_arrays(m, n, d, seed=5)
    P0, Q0 = synth.init_factors(m, n, k, 6)
    dev = Device(0, raise_errors=True)
    res = []
    for split in (0, 1):
        dev.set_factors(P0, Q0)
        dev.set_interactions(data['indptr'], data['indices'], data['ev_ptr'], data['ev_i'])
        dev.set_option('epoch_exact', 1); dev.set_option('chain_split', split)
        t0 = time.perf_counter()
        nll = dev.bpr_epoch(5, 0, 0, 0.02, 0.01, 0.01)[0]
        dt = time.perf_counter() - t0
        P, Q = dev.get_factors()
        res.append((nll, P, Q, dt))
    dev.close()
    print(m, n, d, k, 'equal P %s Q %s nll rel %.1e  ms %.1f / %.1f' % (np.array_equal(res[0][1], res[1][1]), np.array_equal(res[0][2], res[1][2]),
          abs(res[0][0] - res[1][0]) / abs(res[0][0]), 1e3 * res[0][3], 1e3 * res[1][3]), flush=True)
