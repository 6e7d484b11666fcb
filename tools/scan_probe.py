#!/usr/bin/env python3
"""Timing probe of the scoring paths on config 5: python tools/scan_probe.py EPOCHS [users]  (EPOCHS = 0: random factors)
prints kernel ms of yue_topn_scan for the fused kernel and the two-phase path (several chunk growth factors);
PROBE_DEFAULT=1: only the library's default path (for counter passes)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yue_amd import synth
from yue_amd import _shim
from yue_amd._shim import Device
if os.environ.get('YUE_LIB'):
    _shim.LIB_PATH = os.environ['YUE_LIB']
epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 0
m, n, d, k, N = 1000000, 200000, 50, 128, 20
nu = int(sys.argv[2]) if len(sys.argv) > 2 else m
data = synth.make_arrays(m, n, d, seed=20260001)
P0, Q0 = synth.init_factors(m, n, k, 20260002)
if os.environ.get('PROBE_CONSTQ') == 'row':     # timing experiments (diagnostic libraries without sifting): every item row the same / zero
    Q0[:] = Q0[0]
elif os.environ.get('PROBE_CONSTQ') == 'zero':
    Q0[:] = 0
elif os.environ.get('PROBE_CONSTQ') == 'const':
    Q0[:] = 0.01
dev = Device(0, raise_errors=True)
dev.set_factors(P0, Q0)
dev.set_interactions(data['indptr'], data['indices'], data['ev_ptr'], data['ev_i'])
for ep in range(epochs):
    dev.bpr_epoch(20260003, ep, 0, 0.02, 0.01, 0.01)
users = np.arange(nu, dtype=np.int32)
ref = None
configs = [('default (two-phase, growth chosen from the first chunk)', {'scan_filter_ub': int(os.environ['PROBE_FORM'])} if os.environ.get('PROBE_FORM') else {})] if os.environ.get('PROBE_DEFAULT') else None
if os.environ.get('PROBE_UB'):          # one or two blocks of users per filter wave
    configs = [('two-phase, %d stream(s), %d slabs' % (t, sl), {'scan_streams': t, 'scan_slabs': sl}) for t, sl in ((1, 4), (2, 2), (2, 4), (2, 6), (2, 8), (2, 12), (2, 4))]
for label, opts in configs or [('fused', {'scan_two_phase': 0}), ('two-phase x2', {'scan_two_phase': 1, 'scan_growth': 2}), ('two-phase x4', {'scan_two_phase': 1, 'scan_growth': 4}),
                    ('two-phase x8', {'scan_two_phase': 1, 'scan_growth': 8}), ('two-phase x16', {'scan_two_phase': 1, 'scan_growth': 16})]:
    for a, b in opts.items():
        dev.set_option(a, b)
    dev.topn_scan(users[:4096], N)
    t0 = time.perf_counter()
    ids, sc = dev.topn_scan(users, N)
    dt = time.perf_counter() - t0
    ms, events, rescored, bf = dev.scan_stats()
    done, total = dev.scan_work()
    if ref is None:
        ref = (ids, sc)
    same = np.array_equal(ids, ref[0]) and np.array_equal(sc, ref[1])
    print('%d epochs, %d users, %-14s kernel %.1f ms (wall %.1f)  events/user %.1f  exact/user %.1f  tiles %.3f  chunks %d  equal %s' % (
        epochs, nu, label, ms, 1e3 * dt, events / nu, rescored / nu, done / max(1, total), dev.get_option('scan_last_chunks'), same), flush=True)
dev.close()
