"""debug: one small fused epoch, k_round3 vs oracle, with the hot set on / off"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle
from yue_amd import synth
from yue_amd import _shim
from yue_amd._shim import Device
if os.environ.get('YUE_LIB'):
    _shim.LIB_PATH = os.environ['YUE_LIB']
from yue_amd.dist import epoch_round_ptr
orc = oracle.Oracle()
m, n, d, k, W = [int(x) for x in (sys.argv[1:6] if len(sys.argv) > 5 else (300, 400, 20, 10, 128))]
data = synth.make_arrays(m, n, d, seed=5)
P0, Q0 = synth.init_factors(m, n, k, 5)
ev_u = np.repeat(np.arange(m, dtype=np.int32), np.diff(data['ev_ptr']))
rp = np.array(epoch_round_ptr(data['ev_ptr'], W), np.int64)
for lam in (200, 100000, 30):
    dev = Device(0, raise_errors=True)
    dev.set_option('hot_lambda_x100', lam)
    dev.set_factors(P0, Q0)
    dev.set_interactions(data['indptr'], data['indices'], data['ev_ptr'], data['ev_i'])
    Po, Qo = P0.copy(), Q0.copy()
    j = orc.sample_counter(987654321, 0, ev_u, n, data['indptr'], data['indices'])
    nll, sp, sq = dev.bpr_epoch(987654321, 0, W, 0.02, 0.01, 0.01)
    nll_o = orc.bpr_rounds(Po, Qo, ev_u, data['ev_i'], j, rp, 0.02, 0.01, 0.01)
    P, Q = dev.get_factors()
    eP = np.abs(P - Po).max(axis=1); eQ = np.abs(Q - Qo).max(axis=1)
    print('hot_lambda %6.2f: max|dP| %.3e (rows over 1e-6: %d)  max|dQ| %.3e (rows over 1e-6: %d, first %s)  nll %.6f vs %.6f'
          % (lam / 100, eP.max(), (eP > 1e-6).sum(), eQ.max(), (eQ > 1e-6).sum(), np.nonzero(eQ > 1e-6)[0][:10], nll, nll_o))
    dev.close()
