"""CPU: oracle/numpy_cune.py (the NumPy restatement of CUNE's two-level BPR training loop, reference
recommender/advanced/CUNE.py:120-178) against outputs of the reference's own loop (tests/golden/g9_*, made by
tools/make_goldens.py with a synthetic friends' item set: the Word2Vec stage needs gensim, absent here).
The restatement issues the reference's operations, so everything is compared for equality."""
import numpy as np
import pytest

from test_fism_golden import bold_driver
from util import gj, gz

CASES = ['cune_d2_k20_e2', 'cune_d3_k64_e1']


@pytest.mark.parametrize('tag', CASES)
def test_numpy_cune_lands_on_the_reference(tag):
    from oracle.numpy_cune import cune_epoch
    z, meta = gz('g9_%s.npz' % tag), gj('g9_%s.json' % tag)
    assert meta['dtypes'] == ['float32', 'float32'] and meta['loss_type'] == 'float32'
    iters, s = int(z['iters']), float(z['s'])
    P, Q = z['P0'].copy(), z['Q0'].copy()
    per = len(z['u']) // iters
    lr, last = meta['lr0'], 0
    for ep in range(iters):
        sl = slice(ep * per, (ep + 1) * per)
        loss = cune_epoch(P, Q, z['u'][sl], z['i'][sl], z['kk'][sl], z['j'][sl], s, lr, meta['regU'], meta['regI'])
        assert 'CUNE [1] iteration %d: loss = %.4f, delta_loss = %.5f learning_Rate = %.5f' % (ep + 1, loss, last - loss, lr) == meta['lines'][ep]
        lr = bold_driver(lr, last, loss, ep + 1, meta['lr_max'])
        last = loss
    assert type(loss).__name__ == 'float32'
    assert np.array_equal(P, z['P']) and np.array_equal(Q, z['Q'])
    assert float(loss) == float(z['loss']) and lr == float(z['lRate'])
    # both branches are exercised: users with and without friends' items
    assert (z['kk'] < 0).any() and (z['kk'] >= 0).any()
    # the stream is what the loop's structure says: three steps per training event, users consecutive
    assert np.array_equal(z['u'][:per:3], z['ev_u']) and np.array_equal(z['i'][:per:3], z['ev_i'])
