#!/usr/bin/env python3
"""Distance of the S-round semantics from the reference's sequential loop as a function of the round size W
(DESIGN.md section 3): python tools/deviation_table.py c2|c3 [W ...]  -> one JSON line per W.
One epoch from the initial factors on the device sampler's epoch-0 negatives; the sequential end point comes from the
exact device path (option epoch_exact), which tests/test_gpu_exact.py pins to oracle/bpr_oracle.c."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yue_amd import synth                      # noqa: E402
from yue_amd._shim import Device               # noqa: E402

W = {'c3': (1000000, 200000, 50, 128), 'c2': (100000, 50000, 50, 64), 'tiny': (20000, 5000, 20, 128)}
LR, REG = 0.02, 0.01


def rms(a):
    a = a.astype(np.float64, copy=False).ravel()
    return float(np.sqrt(np.dot(a, a) / a.size))


def main():
    name = sys.argv[1]
    m, n, d, k = W[name]
    data = synth.make_arrays(m, n, d, seed=20260001)
    P0, Q0 = synth.init_factors(m, n, k, 20260002)
    dev = Device(0, raise_errors=True)
    dev.set_factors(P0, Q0)
    dev.set_interactions(data['indptr'], data['indices'], data['ev_ptr'], data['ev_i'])
    for kv in filter(None, os.environ.get('YUE_OPTS', '').split(',')):      # tuning options of the S-round path (timing only)
        dev.set_option(kv.split('=')[0], int(kv.split('=')[1]))
    E = int(data['ev_ptr'][-1])
    rounds = [int(x) for x in sys.argv[2:]] or [8192, 57344, 114688, 172032]
    dev.set_option('epoch_exact', 1)
    nll_e = dev.bpr_epoch(20260003, 0, 0, LR, REG, REG)[0]
    dev.set_option('epoch_exact', 0)
    Pe, Qe = dev.get_factors()
    mvP, mvQ = rms(Pe - P0), rms(Qe - Q0)
    for w in rounds + [0]:
        dev.set_factors(P0, Q0)
        weff = w or dev.default_round_events()
        nll_r = dev.bpr_epoch(20260003, 0, weff, LR, REG, REG)[0]
        Pr, Qr = dev.get_factors()
        t0 = time.perf_counter()
        for ep in range(1, 4):
            dev.bpr_epoch(20260003, ep, weff, LR, REG, REG)
        ms = 1e3 * (time.perf_counter() - t0) / 3
        print(json.dumps({'workload': name, 'W': weff, 'default': w == 0, 'ms_per_epoch': round(ms, 2), 'loss_rel': (nll_r - nll_e) / nll_e,
                          'nll_per_triplet': [nll_e / E, nll_r / E],
                          'rms_distance_over_movement': [rms(Pr - Pe) / mvP, rms(Qr - Qe) / mvQ],
                          'normwise_rel': [float(np.abs(Pr - Pe).max() / np.abs(Pe).max()), float(np.abs(Qr - Qe).max() / np.abs(Qe).max())]}), flush=True)
    dev.close()


if __name__ == '__main__':
    main()
