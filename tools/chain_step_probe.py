#!/usr/bin/env python3
"""Latency of the exact path's two kinds of dependent step, measured on streams built for it (yue_bpr_replay):
  step: R independent runs of L triplets on items nobody else touches -- time / L = one in-run step (no cross-wave hand-off);
  hop:  H runs of ONE triplet that all touch item 0 -- time / H = one cross-wave hand-off of a row.
python tools/chain_step_probe.py [NAME=VALUE ...]   (options for yue_set_option, e.g. chain_split=1 chain_fast=1)"""
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


def timed_replay(dev, P0, Q0, u, i, j):
    """seconds of the dataflow launch alone (HIP events inside the library: read-only option chain_last_us), best of three"""
    best = 1e9
    for _ in range(3):
        dev.set_factors(P0, Q0)
        dev.bpr_replay(u, i, j, 0.02, 0.01, 0.01)
        best = min(best, 1e-6 * dev.get_option('chain_last_us'))
    return best


def main():
    k = 128
    R, POOL = 256, int(os.environ.get('PROBE_POOL', '4096'))    # a run re-visits an item of its private pool POOL / 2 triplets later (>= 64: beyond the prefetch ring)
    n = R * POOL + 400000
    m = 300000
    P0, Q0 = synth.init_factors(m, n, k, 3)
    dev = Device(0, raise_errors=True)
    for kv in sys.argv[1:]:
        a, b = kv.split('=')
        dev.set_option(a, int(b))
    rs = np.random.RandomState(5)
    res = {}
    for L in (4000, 20000):
        u = np.repeat(np.arange(R, dtype=np.int32), L)
        t = np.tile(np.arange(L, dtype=np.int64), R)
        base = np.repeat(np.arange(R, dtype=np.int64) * POOL, L)
        i = (base + (2 * t) % POOL).astype(np.int32)
        j = (base + (2 * t + 1) % POOL).astype(np.int32)
        res[L] = timed_replay(dev, P0, Q0, u, i, j)
    step = (res[20000] - res[4000]) / 16000
    print('pool %d: in-run step: %.3f us (dataflow launch alone)  (%d runs; %d triplets per run: %.1f ms, %d: %.1f ms)' % (POOL, 1e6 * step, R, 4000, 1e3 * res[4000], 20000, 1e3 * res[20000]), flush=True)
    if os.environ.get('PROBE_STEP_ONLY'):
        dev.close()
        return
    for H in (100000, 300000):
        u = np.arange(H, dtype=np.int32)
        i = np.zeros(H, np.int32) + (n - 1)
        j = (R * POOL + rs.permutation(390000)[:H]).astype(np.int32)
        res[H] = timed_replay(dev, P0, Q0, u, i, j)
    hop = (res[300000] - res[100000]) / 200000
    print('cross-wave hop: %.3f us (dataflow launch alone)  (%d single-triplet runs on one item: %.1f ms, %d: %.1f ms)' % (1e6 * hop, 100000, 1e3 * res[100000], 300000, 1e3 * res[300000]), flush=True)
    dev.close()


if __name__ == '__main__':
    main()
