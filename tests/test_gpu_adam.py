"""GPU: the reference's live BPR path (recommender/cf/BPR.py:83-129, a TensorFlow-1 graph: softplus loss + l2 terms, Adam)
through the C ABI (yue_adam_step) against oracle/numpy_adam.py.  PARITY UNPINNED against the reference itself: TensorFlow
cannot be installed here, the oracle restates the graph as written; what these tests pin is that the device
evaluates that restatement (float32, atomics reorder the gradient sums -> 1e-5 rel)."""
import numpy as np
import pytest

from util import rel_err

pytestmark = pytest.mark.gpu


def _batch(rs, m, n, events, negs):
    u = np.repeat(rs.randint(0, m, size=events), negs).astype(np.int32)
    i = np.repeat(rs.randint(0, n, size=events), negs).astype(np.int32)
    j = rs.randint(0, n, size=events * negs).astype(np.int32)
    return u, i, j


@pytest.mark.parametrize('m,n,k,events,negs', [(300, 500, 10, 64, 100), (2000, 3000, 128, 512, 100), (50, 80, 200, 7, 3)])
def test_steps_match_the_numpy_restatement(m, n, k, events, negs):
    from oracle.numpy_adam import adam_step, new_state, truncated_normal
    from yue_amd._shim import Device
    rs = np.random.RandomState(k)
    U0, V0 = truncated_normal(rs, (m, k), 0.005), truncated_normal(rs, (n, k), 0.005)
    assert np.abs(U0).max() <= 0.01 and U0.dtype == np.float32
    dev = Device(0, raise_errors=True)
    dev.set_factors(U0, V0)
    dev.adam_reset()
    U, V = U0.copy(), V0.copy()
    st = new_state(U, V)
    for step in range(1, 6):
        u, i, j = _batch(rs, m, n, events, negs)
        loss = dev.adam_step(u, i, j, 0.02, 0.01, step)
        loss_o = adam_step(U, V, st, u, i, j, 0.02, 0.01, step)
        assert abs(loss - loss_o) < 1e-5 * abs(loss_o)
    P, Q = dev.get_factors()
    # Adam divides by sqrt(v): where a summed gradient is close to zero, the float32 order of the sum (atomics on the
    # device) decides about a visible part of the step -- a few elements differ at the 1e-3 level, the bulk at 1e-6 .. 1e-5
    for got, want in ((P, U), (Q, V)):
        err = np.abs(got.astype(np.float64) - want) / np.abs(want).max()
        assert err.max() < 5e-3 and np.quantile(err, 0.99) < 1e-4 and np.median(err) < 1e-5
    # rows nobody touched in the last step still move (TF-1's sparse Adam applies m / (sqrt(v) + eps) to every row)
    untouched = np.setdiff1d(np.arange(n), np.concatenate([i, j]))
    assert len(untouched) == 0 or not np.array_equal(Q[untouched], V0[untouched])
    dev.close()


def test_live_path_through_the_plugin_surface(tmp_path, capsys):
    import random
    from yue_amd.recommender.cf.BPR import BPR
    from test_host_golden import _c1_conf, _load
    conf = _c1_conf(tmp_path, 10, 3, '5,10')
    conf.config['bpr.hip'] = '-mode adam -gpu 0'
    rec = BPR(conf, _load(conf), [])
    rec.readConfiguration()
    random.seed(5)
    np.random.seed(5)
    rec.initModel()
    assert np.abs(rec.P).max() <= 0.01 and np.abs(rec.Q).max() <= 0.01
    capsys.readouterr()
    rec.buildModel()
    out = capsys.readouterr().out
    losses = [float(ln.split('loss:')[1]) for ln in out.splitlines() if ln.startswith('iteration:')]
    assert len(losses) == 3 and losses[2] < losses[0] and abs(losses[0] - 51200 * np.log(2)) < 0.01 * 51200 * np.log(2)
    assert out.count('rank measure...') == 3                      # ranking_performance after every step (:129)
    # the other paths still work on the same device context afterwards (gradient buffers clean)
    conf.config['bpr.hip'] = '-mode epoch -round 2048 -seed 5 -gpu 0'
    rec2 = BPR(conf, _load(conf), [])
    rec2.readConfiguration()
    np.random.seed(5)
    rec2.initModel()
    rec2.buildModel()
    assert 'iteration' in capsys.readouterr().out
