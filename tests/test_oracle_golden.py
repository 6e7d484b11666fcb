"""Pins oracle/ (our CPU restatement) to outputs of the reference itself (tests/golden, made by
tools/make_goldens.py from /root/reference).  CPU only."""
import numpy as np
import pytest

from yue_amd import synth
from util import csr_from_events, gj, gz, mask_rows, rel_err, sha

CASES = ['c1_k10_e1', 'c1_k10_e5', 'd2_k64_e1', 'd3_k128_e2']


def _case(tag):
    z, meta = gz('g4_%s.npz' % tag), gj('g4_%s.json' % tag)
    iters = int(z['iters'])
    E = len(z['u']) // iters
    return z, meta, iters, E


@pytest.mark.parametrize('tag', CASES)
def test_python_sampler_replica_is_bit_exact(orc, tag):
    # recommender/cf/BPR.py:46-48 driven by random.seed(seed): every raw draw and every accepted j
    z, _, iters, E = _case(tag)
    m, n = int(z['m']), int(z['n'])
    indptr, indices = csr_from_events(z['u'][:E], z['i'][:E], m)
    j, draws = orc.sample_python(int(z['seed']), iters, z['u'][:E], n, indptr, indices, want_draws=True)
    assert np.array_equal(draws, z['draws'])
    assert np.array_equal(j, z['j'])


@pytest.mark.parametrize('tag', CASES)
def test_sequential_epochs_match_reference(orc, tag):
    # BPR.py:40-62 + IterativeRecommender.py:47-75: factors within 1e-5 rel, printed lines identical
    z, meta, iters, E = _case(tag)
    m, n, k, seed = int(z['m']), int(z['n']), int(z['k']), int(z['seed'])
    P, Q = synth.init_factors(m, n, k, seed)
    assert sha(P) == meta['P0_sha256'] and sha(Q) == meta['Q0_sha256']
    lr, last = 0.02, 0.0
    for ep in range(iters):
        sl = slice(ep * E, (ep + 1) * E)
        nll = orc.bpr_sequential(P, Q, z['u'][sl], z['i'][sl], z['j'][sl], lr, 0.01, 0.01)
        # BPR.py:59 under NumPy 2 promotion: python-float * float32 -> float32, loss becomes float32
        reg = np.float32(0.01) * np.float32(orc.sumsq(P)) + np.float32(0.01) * np.float32(orc.sumsq(Q))
        loss = np.float32(nll) + reg
        line = 'BPR [1] iteration %d: loss = %.4f, delta_loss = %.5f learning_Rate = %.5f' % (ep + 1, loss, last - loss, lr)
        assert line == meta['lines'][ep]
        if not abs(last - loss) < 1e-3:
            if ep + 1 > 1:
                lr = lr * 1.01 if abs(last) > abs(loss) else lr * 0.5
            lr = min(lr, 1.0)
        last = loss
    assert rel_err(P, z['P']) < 1e-5 and rel_err(Q, z['Q']) < 1e-5
    assert abs(float(last) - float(z['loss'])) <= 1e-5 * abs(float(z['loss']))
    assert abs(lr - float(z['lRate'])) < 1e-12


@pytest.mark.parametrize('tag', ['c1_k10_e1', 'd2_k64_e1', 'd3_k128_e2'])
def test_numpy_restatement_lands_on_the_reference_bit_for_bit(orc, tag):
    # oracle/numpy_loop.py issues the same NumPy operations as BPR.py:50-58 -> identical factors on this NumPy
    # build; the C oracle (own dot order) stays within 1e-6 of it.  (Learning rate: unchanged after iteration 1.)
    from oracle.numpy_loop import bpr_loop
    z, meta, iters, E = _case(tag)
    m, n, k, seed = int(z['m']), int(z['n']), int(z['k']), int(z['seed'])
    P, Q = synth.init_factors(m, n, k, seed)
    Pc, Qc = P.copy(), Q.copy()
    for ep in range(iters):
        sl = slice(ep * E, (ep + 1) * E)
        nll = bpr_loop(P, Q, z['u'][sl], z['i'][sl], z['j'][sl], 0.02, 0.01, 0.01)
        nll_c = orc.bpr_sequential(Pc, Qc, z['u'][sl], z['i'][sl], z['j'][sl], 0.02, 0.01, 0.01)
        assert abs(nll - nll_c) <= 1e-6 * abs(nll)
    assert np.array_equal(P, z['P']) and np.array_equal(Q, z['Q'])
    assert rel_err(Pc, P) < 1e-6 and rel_err(Qc, Q) < 1e-6


def test_hogwild_baseline_with_one_thread_is_the_sequential_loop(orc):
    z, _, iters, E = _case('d2_k64_e1')
    m, n, k, seed = int(z['m']), int(z['n']), int(z['k']), int(z['seed'])
    P, Q = synth.init_factors(m, n, k, seed)
    P2, Q2 = P.copy(), Q.copy()
    a = orc.bpr_sequential(P, Q, z['u'][:E], z['i'][:E], z['j'][:E], 0.02, 0.01, 0.01)
    b = orc.bpr_hogwild(P2, Q2, z['u'][:E], z['i'][:E], z['j'][:E], 0.02, 0.01, 0.01, 1)
    assert a == b and np.array_equal(P, P2) and np.array_equal(Q, Q2)
    P3, Q3 = synth.init_factors(m, n, k, seed)
    c = orc.bpr_hogwild(P3, Q3, z['u'][:E], z['i'][:E], z['j'][:E], 0.02, 0.01, 0.01, 4)     # races: close, not equal
    assert abs(c - a) < 0.05 * abs(a) and np.isfinite(P3).all() and np.isfinite(Q3).all()


@pytest.mark.parametrize('tag,N', [('c1_top10', 10), ('c1_top20', 20)])
def test_overwrite_scan_lists_match_reference(orc, tag, N):
    # IterativeRecommender.py:93-145 on the reference's own trained factors: integer lists identical
    z = gz('g4_c1_k10_e1.npz')
    ev = gz('g2_events_c1.npz')
    g = gz('g5_%s.npz' % tag)
    indptr, indices = csr_from_events(ev['ev_u'], ev['ev_i'], int(z['m']))
    users = g['test_users']
    mp, mi = mask_rows(indptr, indices, users)
    ids, sc, rc = orc.topn_scan(z['P'], z['Q'], users, N, mp, mi)
    assert rc == 0
    assert np.array_equal(ids, g['rec_ids'])
    # F4: it is not a top-N -- slot 0 is the maximum, the rest is order dependent
    tid, tsc, _ = orc.topn_true(z['P'], z['Q'], users, N, mp, mi)
    assert np.array_equal(ids[:, 0], tid[:, 0])
    assert (ids != tid).any()
    for t in range(16):
        s = orc.scores(z['P'], z['Q'], int(users[t]))
        assert np.abs(s - g['predict16'][t]).max() <= 4e-7 * np.abs(g['predict16'][t]).max()


def test_scan_too_few_candidates_flags(orc):
    # IterativeRecommender.py:126 raises IndexError when fewer than N candidates remain
    rs = np.random.RandomState(0)
    P = rs.rand(2, 4).astype(np.float32)
    Q = rs.rand(6, 4).astype(np.float32)
    mp = np.array([0, 3, 3], np.int64)
    mi = np.array([0, 2, 4], np.int32)
    ids, sc, rc = orc.topn_scan(P, Q, np.array([0, 1], np.int32), 5, mp, mi)
    assert rc == -1
    assert set(ids[1].tolist()) <= set(range(6)) and (ids[0][3:] == -1).all()


def test_rounds_of_one_equal_sequential(orc):
    z, _, _, E = _case('d2_k64_e1')
    m, n, k = int(z['m']), int(z['n']), int(z['k'])
    P1, Q1 = synth.init_factors(m, n, k, 7)
    P2, Q2 = P1.copy(), Q1.copy()
    T = 2000
    a = orc.bpr_sequential(P1, Q1, z['u'][:T], z['i'][:T], z['j'][:T], 0.02, 0.01, 0.01)
    b = orc.bpr_rounds(P2, Q2, z['u'][:T], z['i'][:T], z['j'][:T], np.arange(T + 1), 0.02, 0.01, 0.01)
    assert rel_err(P2, P1) < 2e-6 and rel_err(Q2, Q1) < 2e-6
    assert abs(a - b) < 1e-9 * abs(a)


def test_rounds_with_sequential_user_rows_are_pinned_to_the_sequential_loop(orc):
    """orc_bpr_rounds_seq_user (round 4: user rows sequential, item rows under round semantics): a round of ONE triplet is the
    reference's loop on the reference's own triplet stream (up to the x + (x' - x) rounding of the two item rows); with whole users
    per round the user rows only differ from the sequential loop's through the staleness of the item rows -- and a round that holds
    every item at most once IS the sequential loop."""
    z, _, _, E = _case('d2_k64_e1')
    m, n, k = int(z['m']), int(z['n']), int(z['k'])
    P1, Q1 = synth.init_factors(m, n, k, 7)
    P2, Q2 = P1.copy(), Q1.copy()
    T = 2000
    a = orc.bpr_sequential(P1, Q1, z['u'][:T], z['i'][:T], z['j'][:T], 0.02, 0.01, 0.01)
    b = orc.bpr_rounds_seq_user(P2, Q2, z['u'][:T], z['i'][:T], z['j'][:T], np.arange(T + 1), 0.02, 0.01, 0.01)
    assert rel_err(P2, P1) < 2e-6 and rel_err(Q2, Q1) < 2e-6 and abs(a - b) < 1e-9 * abs(a)
    # one user, every item touched once, all in ONE round: no item row is read after it was written -> the sequential loop again
    rs = np.random.RandomState(1)
    items = rs.permutation(n)[:200].astype(np.int32)
    u = np.zeros(100, np.int32)
    P3, Q3 = synth.init_factors(m, n, k, 9)
    P4, Q4 = P3.copy(), Q3.copy()
    a = orc.bpr_sequential(P3, Q3, u, items[:100], items[100:], 0.02, 0.01, 0.01)
    b = orc.bpr_rounds_seq_user(P4, Q4, u, items[:100], items[100:], np.array([0, 100], np.int64), 0.02, 0.01, 0.01)
    assert np.array_equal(P4, P3) and rel_err(Q4, Q3) < 2e-6 and abs(a - b) < 1e-12 * abs(a)
    # and the plain S-round (user row summed over the round) is NOT: the difference the round-4 form removes
    P5, Q5 = synth.init_factors(m, n, k, 9)
    orc.bpr_rounds(P5, Q5, u, items[:100], items[100:], np.array([0, 100], np.int64), 0.02, 0.01, 0.01)
    assert rel_err(P5, P3) > 1e-4


def test_dataflow_timing_model(orc):
    """orc_dataflow_model (a timing model of the exact path's launch, not a checker): hand-checkable cases."""
    u = np.array([0, 0, 0, 1, 1, 2], np.int32)
    i = np.array([0, 1, 2, 3, 4, 0], np.int32)
    j = np.array([5, 6, 7, 8, 9, 6], np.int32)
    # no shared rows except run 2 (user 2) touching items 0 and 6 of run 0: independent runs in parallel
    t, hops = orc.dataflow_model(u, i, j, 10, 8, 1.0, 0.5, 2.0, 2.0)
    # run 0: 1 + 3 * 0.5 = 2.5; run 2 starts at 1.0 but needs item 0 (ready 1.5 + 2) and item 6 (ready 2.0 + 2): 4.0 + 0.5
    assert abs(t - 4.5) < 1e-12 and hops == 1
    # one worker: the runs queue up (run 1 starts when run 0 is done)
    t1, _ = orc.dataflow_model(u, i, j, 10, 1, 1.0, 0.5, 2.0, 2.0)
    assert abs(t1 - (2.5 + 1.0 + 1.0 + 1.0 + 0.5)) < 1e-12
    # a skipped triplet costs nothing
    j2 = j.copy(); j2[1] = -1
    t2, _ = orc.dataflow_model(u, i, j2, 10, 8, 1.0, 0.5, 0.0, 0.0)
    assert abs(t2 - 2.0) < 1e-12


def test_counter_sampler_properties(orc):
    ev = gz('g2_events_c1.npz')
    m, n = 1000, 1000
    indptr, indices = csr_from_events(ev['ev_u'], ev['ev_i'], m)
    j0 = orc.sample_counter(123, 0, ev['ev_u'], n, indptr, indices)
    j1 = orc.sample_counter(123, 1, ev['ev_u'], n, indptr, indices)
    assert (j0 >= 0).all() and (j0 < n).all() and (j0 != j1).mean() > 0.9
    for e in range(0, len(j0), 97):
        u = ev['ev_u'][e]
        assert j0[e] not in set(indices[indptr[u]:indptr[u + 1]].tolist())
    # an offset slice draws the same values as the full call (event index is the counter)
    part = orc.sample_counter(123, 0, ev['ev_u'][500:900], n, indptr, indices, e0=500)
    assert np.array_equal(part, j0[500:900])
    # roughly uniform over the non-listened items
    hist = np.bincount(j0, minlength=n)
    assert hist.max() < 60
