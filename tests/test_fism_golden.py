"""CPU: oracle/numpy_fism.py (the NumPy restatement of the reference's FISM) against outputs of the reference
itself (tests/golden/g8_*, made by tools/make_goldens.py).  The restatement issues the reference's NumPy
operations, so everything is compared for equality."""
import numpy as np
import pytest

from util import gj, gz

CASES = ['fism_c1_k10_e2', 'fism_d2_k64_e1', 'fism_d3_k130_e3']
LR0, REG = 0.015, 0.01            # FISM.conf: learnRate -init 0.015, reg.lambda -u/-i/-b 0.01


def fism_case(tag):
    z, meta = gz('g8_%s.npz' % tag), gj('g8_%s.json' % tag)
    m = int(z['m'])
    ptr = np.zeros(m + 1, np.int64)
    np.add.at(ptr, z['ev_u'].astype(np.int64) + 1, 1)
    return z, meta, np.cumsum(ptr)


def bold_driver(lr, last, loss, iteration, max_lr=1.0):
    # base/IterativeRecommender.py:47-75 (the lines FISM inherits)
    if not abs(last - loss) < 1e-3:
        if iteration > 1:
            lr = lr * 1.01 if abs(last) > abs(loss) else lr * 0.5
        lr = min(lr, max_lr)
    return lr


@pytest.mark.parametrize('tag', CASES)
def test_numpy_fism_lands_on_the_reference(tag):
    from oracle.numpy_fism import fism_epoch, fism_regulariser, fism_scores, overwrite_scan
    z, meta, ptr = fism_case(tag)
    assert meta['dtypes'] == ['float64', 'float32', 'float64']
    iters, rho, alpha = int(z['iters']), int(z['rho']), float(z['alpha'])
    P, Q, Bi = z['P0'].copy(), z['Q0'].copy(), z['B0'].copy()
    per = len(z['negs']) // iters
    lr, last = LR0, 0
    for ep in range(iters):
        half = fism_epoch(P, Q, Bi, ptr, z['ev_i'], z['negs'][ep * per:(ep + 1) * per], rho, alpha, lr, REG, REG)
        loss = half + fism_regulariser(P, Q, Bi, REG, REG, REG)
        assert 'FISM [1] iteration %d: loss = %.4f, delta_loss = %.5f learning_Rate = %.5f' % (ep + 1, loss, last - loss, lr) == meta['lines'][ep]
        lr = bold_driver(lr, last, loss, ep + 1)
        last = loss
    assert np.array_equal(P, z['P']) and np.array_equal(Q, z['Q']) and np.array_equal(Bi, z['Bi'])
    assert loss == float(z['loss']) and lr == float(z['lRate'])
    tu, N = z['test_users'], z['rec_ids'].shape[1]
    for t in range(16):
        u = int(tu[t])
        assert np.array_equal(fism_scores(P, Q, Bi, z['ev_i'][ptr[u]:ptr[u + 1]]), z['predict16'][t])
    for t in range(0, len(tu), 7):
        items = z['ev_i'][ptr[int(tu[t])]:ptr[int(tu[t]) + 1]]
        assert overwrite_scan(fism_scores(P, Q, Bi, items), items, N)[0] == list(z['rec_ids'][t])


def test_fism_negatives_follow_the_draw_log():
    # the accepted negatives are the logged draws minus the rejected ones (FISM.py:50-53), users with one event skipped
    z, _, ptr = fism_case('fism_d2_k64_e1')
    rho = int(z['rho'])
    it = iter(z['draws'].tolist())
    got = []
    for u in range(len(ptr) - 1):
        items = set(z['ev_i'][ptr[u]:ptr[u + 1]].tolist())
        if ptr[u + 1] - ptr[u] == 1:
            continue
        for _ in range((ptr[u + 1] - ptr[u]) * rho):
            j = next(it)
            while j in items:
                j = next(it)
            got.append(j)
    assert got == z['negs'].tolist() and next(it, None) is None


def test_rounds_of_one_user_are_the_reference_pass():
    # oracle/numpy_fism.py: fism_rounds with one user per round against the reference's own output (pins the round oracle)
    from oracle.numpy_fism import fism_rounds
    z, meta, ptr = fism_case('fism_c1_k10_e2')
    iters, rho, alpha = int(z['iters']), int(z['rho']), float(z['alpha'])
    P, Q, Bi = z['P0'].copy(), z['Q0'].copy(), z['B0'].copy()
    per = len(z['negs']) // iters
    half = fism_rounds(P, Q, Bi, ptr, z['ev_i'], z['negs'][:per], rho, alpha, LR0, REG, REG, 1)
    from oracle.numpy_fism import fism_epoch
    Pe, Qe, Be = z['P0'].copy(), z['Q0'].copy(), z['B0'].copy()
    half_e = fism_epoch(Pe, Qe, Be, ptr, z['ev_i'], z['negs'][:per], rho, alpha, LR0, REG, REG)
    # x + (x' - x) may differ from x' in the last float32 bit of a Q element; what later users read from it moves the
    # float64 arrays by as much
    assert np.abs(P - Pe).max() <= 1e-6 * np.abs(Pe).max() and np.abs(Bi - Be).max() <= 1e-6 * np.abs(Be).max()
    assert np.abs(Q - Qe).max() <= 1e-6 * np.abs(Qe).max() and abs(half - half_e) <= 1e-6 * half_e
    # larger rounds move away from the sequential pass, but only slightly on this data
    P8, Q8, B8 = z['P0'].copy(), z['Q0'].copy(), z['B0'].copy()
    half8 = fism_rounds(P8, Q8, B8, ptr, z['ev_i'], z['negs'][:per], rho, alpha, LR0, REG, REG, 8)
    assert abs(half8 - half_e) < 0.02 * half_e and np.abs(Q8 - Qe).max() > 0
