"""Where the time of one round launch goes (diagnostic; needs `make -C yue_amd/csrc stamps`).

Loads the stamps build of the library, runs one C3-shaped epoch and prints, for a mid-epoch launch of
k_round, the mean per-wave duration of each phase and the launch's overall timeline.
usage: python tools/round_stamps.py [--users 1000000] [--items 200000] [--k 128] [--round 32768] [--stage 1]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yue_amd import _shim, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--users', type=int, default=1000000)
ap.add_argument('--items', type=int, default=200000)
ap.add_argument('--d', type=int, default=50)
ap.add_argument('--k', type=int, default=128)
ap.add_argument('--round', type=int, default=0, help='round size (0 = the library default)')
ap.add_argument('--stage', type=int, default=1)
ap.add_argument('--tpw', type=int, default=0)
ap.add_argument('--launch', type=int, default=150, help='which round launch of the second epoch to stamp')
args = ap.parse_args()

_shim.LIB_PATH = os.path.join(ROOT, 'yue_amd', 'csrc', 'libyue_hip_stamps.so')
dev = _shim.Device(0, raise_errors=True)
data = synth.make_arrays(args.users, args.items, args.d)
P, Q = synth.init_factors(args.users, args.items, args.k)
dev.set_factors(P, Q)
dev.set_interactions(data['indptr'], data['indices'], data['ev_ptr'], data['ev_i'])
dev.set_option('round_stage', args.stage)
if args.tpw:
    dev.set_option('round_tpw', args.tpw)
dev.bpr_epoch(1, 0, args.round, 0.02, 0.01, 0.01)          # warm
dev.set_option('debug_stamp_launch', args.launch)
dev.bpr_epoch(1, 1, args.round, 0.02, 0.01, 0.01)
lib = _shim.load_library()
cap = 1 << 17
buf = np.zeros((cap, 8), np.uint64)
nw = C.c_int64()
lib.yue_debug_get_stamps(dev._ctx, buf.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_int64(cap), C.byref(nw))
st = buf[:nw.value].astype(np.float64) * 0.01               # 100 MHz -> microseconds
t0 = st[:, 0].min()
names = ['header (scalar loads)', 'row gathers', 'dots + sigmoids', 'issue of stores / atomics', 'drain own stores (k_round)',
         'count decrements (k_round)', 'last-toucher rewrites (k_round) / drain of the stores (k_round_m)']
print('waves %d   launch span (first start .. last end) %.1f us' % (nw.value, st[:, 7].max() - t0))
print('wave start  : mean %.1f  p50 %.1f  p99 %.1f  max %.1f us after the first' % (
    (st[:, 0] - t0).mean(), np.percentile(st[:, 0] - t0, 50), np.percentile(st[:, 0] - t0, 99), (st[:, 0] - t0).max()))
for q, name in enumerate(names):
    dlt = st[:, q + 1] - st[:, q]
    print('%-28s mean %6.2f  p50 %6.2f  p99 %6.2f us' % (name, dlt.mean(), np.percentile(dlt, 50), np.percentile(dlt, 99)))
life = st[:, 7] - st[:, 0]
print('%-28s mean %6.2f  p50 %6.2f  p99 %6.2f us' % ('wave lifetime', life.mean(), np.percentile(life, 50), np.percentile(life, 99)))
for q in range(8):
    print('phase boundary %d reached: first %.1f  mean %.1f  last %.1f us' % (q, st[:, q].min() - t0, st[:, q].mean() - t0, st[:, q].max() - t0))
blk = np.arange(nw.value) // 4
print('mean wave lifetime by (block % 8), i.e. by XCD:', ' '.join('%.1f' % life[(blk % 8) == x].mean() for x in range(8)))
print('mean wave end by grid decile:', ' '.join('%.1f' % (st[c:c + nw.value // 10, 7].mean() - t0) for c in range(0, nw.value - nw.value // 10 + 1, nw.value // 10)))
slow = np.argsort(-life)[:200]
print('slowest 200 waves: mean phases', ' '.join('%.1f' % (st[slow, q + 1] - st[slow, q]).mean() for q in range(7)))
