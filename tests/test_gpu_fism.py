"""GPU parity of the FISM path, through the C ABI: the device runs the reference's
sequential epoch; compared with what the reference produced (tests/golden/g8_*) and with the NumPy oracle.
Tolerances: the reference's BLAS dot order is not pinned, ours is a 64-lane butterfly in double ->
float64 arrays within 1e-11 rel, the float32 Q within 1e-6 rel; printed lines and integer lists equal."""
import glob
import random

import numpy as np
import pytest

from test_fism_golden import CASES, LR0, REG, bold_driver, fism_case
from test_host_golden import _conf_text, _load
from util import gj, gz, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    from yue_amd._shim import Device
    d = Device(0, raise_errors=True)
    yield d
    d.close()


def coefs(ptr, alpha):
    return np.array([pow(int(nu) - 1, -alpha) if nu > 1 else 0.0 for nu in np.diff(ptr)], np.float64)


@pytest.mark.parametrize('tag', CASES)
def test_epochs_match_reference(dev, tag):
    z, meta, ptr = fism_case(tag)
    iters, rho, alpha = int(z['iters']), int(z['rho']), float(z['alpha'])
    dev.fism_set_model(z['P0'], z['Q0'], z['B0'])
    per = len(z['negs']) // iters
    lr, last = LR0, 0
    for ep in range(iters):
        half, sp, sq, sb = dev.fism_epoch(ptr, z['ev_i'], z['negs'][ep * per:(ep + 1) * per], rho, coefs(ptr, alpha), lr, REG, REG)
        loss = np.float64(half) + (REG * np.float64(sp) + REG * np.float32(sq) + REG * np.float64(sb))
        assert 'FISM [1] iteration %d: loss = %.4f, delta_loss = %.5f learning_Rate = %.5f' % (ep + 1, loss, last - loss, lr) == meta['lines'][ep]
        lr = bold_driver(lr, last, loss, ep + 1)
        last = loss
    P, Q, Bi = np.empty_like(z['P']), np.empty_like(z['Q']), np.empty_like(z['Bi'])
    dev.fism_get_model(P, Q, Bi)
    assert rel_err(P, z['P']) < 1e-11 and rel_err(Bi, z['Bi']) < 1e-11 and rel_err(Q, z['Q']) < 1e-6
    assert abs(loss - float(z['loss'])) < 1e-9 * abs(float(z['loss'])) and lr == float(z['lRate'])
    # predict + selection on the trained model
    tu, N = z['test_users'], z['rec_ids'].shape[1]
    for t in range(16):
        u = int(tu[t])
        s = dev.fism_scores(z['ev_i'][ptr[u]:ptr[u + 1]])
        assert np.abs(s - z['predict16'][t]).max() < 1e-12 * np.abs(z['predict16'][t]).max() + 1e-15
    rows = [z['ev_i'][ptr[int(u)]:ptr[int(u) + 1]] for u in tu]
    rp = np.zeros(len(rows) + 1, np.int64)
    rp[1:] = np.cumsum([len(r) for r in rows])
    ids, sc = dev.fism_topn_scan(rp, np.concatenate(rows), N)
    assert np.array_equal(ids, z['rec_ids'])
    assert (np.diff(sc, axis=1) <= 0).sum() >= 0            # scores are the slots' values (not necessarily sorted: overwrite-scan)


def test_against_the_numpy_oracle_on_other_inputs(dev):
    # ragged users (0, 1 and many events), duplicates inside a user, rho = 3, k = 200
    from oracle.numpy_fism import fism_epoch, fism_scores, overwrite_scan
    rng = np.random.RandomState(5)
    n, k, rho, alpha = 90, 200, 3, 0.6
    sizes = [0, 1, 5, 12, 1, 30, 2, 0, 7]
    ptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    ev_i = np.concatenate([rng.randint(0, n // 2, s) for s in sizes]).astype(np.int32)
    negs = []
    for u, s in enumerate(sizes):
        if s > 1:
            mine = set(ev_i[ptr[u]:ptr[u + 1]].tolist())
            negs += [int(rng.choice([x for x in range(n) if x not in mine])) for _ in range(s * rho)]
    negs = np.array(negs, np.int32)
    P0, Q0, B0 = rng.rand(n, k) / 100, (rng.rand(n, k) / 10).astype(np.float32), rng.rand(n) / 100
    dev.fism_set_model(P0, Q0, B0)
    half, sp, sq, sb = dev.fism_epoch(ptr, ev_i, negs, rho, coefs(ptr, alpha), 0.02, 0.01, 0.03)
    Po, Qo, Bo = P0.copy(), Q0.copy(), B0.copy()
    half_o = fism_epoch(Po, Qo, Bo, ptr, ev_i, negs, rho, alpha, 0.02, 0.01, 0.03)
    P, Q, Bi = np.empty_like(P0), np.empty_like(Q0), np.empty_like(B0)
    dev.fism_get_model(P, Q, Bi)
    assert rel_err(P, Po) < 1e-11 and rel_err(Bi, Bo) < 1e-11 and rel_err(Q, Qo) < 1e-6 and abs(half - half_o) < 1e-10 * half_o
    assert abs(sp - (Po * Po).sum()) < 1e-10 * sp and abs(sb - Bo.dot(Bo)) < 1e-10 * sb and abs(sq - float((Qo.astype(np.float64) ** 2).sum())) < 1e-6 * sq
    users = [2, 3, 5, 8]
    rows = [ev_i[ptr[u]:ptr[u + 1]] for u in users]
    rp = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
    ids, _ = dev.fism_topn_scan(rp, np.concatenate(rows), 10)
    for t, r in enumerate(rows):
        assert overwrite_scan(fism_scores(P, Q, Bi, r), r, 10)[0] == ids[t].tolist()
    with pytest.raises(IndexError):                      # fewer than N candidates (reference: IndexError at :126)
        dev.fism_topn_scan(np.array([0, n - 3], np.int64), np.arange(n - 3, dtype=np.int32), 10)
    from yue_amd._shim import YueHipError
    with pytest.raises(YueHipError):                     # wrong number of negatives
        dev.fism_epoch(ptr, ev_i, negs[:-1], rho, coefs(ptr, alpha), 0.02, 0.01, 0.03)


@pytest.mark.parametrize('tag,round_users', [('fism_c1_k10_e2', 1), ('fism_c1_k10_e2', 16), ('fism_d2_k64_e1', 64), ('fism_d3_k130_e3', 1000)])
def test_rounds_match_the_numpy_oracle(dev, tag, round_users):
    # throughput form: rounds of users (k_fism_round, one wave per user) against oracle/numpy_fism.py: fism_rounds;
    # one user per round also against the reference's own result
    from oracle.numpy_fism import fism_rounds
    z, meta, ptr = fism_case(tag)
    iters, rho, alpha = int(z['iters']), int(z['rho']), float(z['alpha'])
    per = len(z['negs']) // iters
    dev.fism_set_model(z['P0'], z['Q0'], z['B0'])
    half, sp, sq, sb = dev.fism_rounds(ptr, z['ev_i'], z['negs'][:per], rho, coefs(ptr, alpha), round_users, LR0, REG, REG)
    Po, Qo, Bo = z['P0'].copy(), z['Q0'].copy(), z['B0'].copy()
    half_o = fism_rounds(Po, Qo, Bo, ptr, z['ev_i'], z['negs'][:per], rho, alpha, LR0, REG, REG, round_users)
    P, Q, Bi = np.empty_like(Po), np.empty_like(Qo), np.empty_like(Bo)
    dev.fism_get_model(P, Q, Bi)
    # the float32 sums of a round's differences come in atomic order: Q within float32 rounding, and the float64 arrays
    # that later users compute from Q move by as much
    assert rel_err(P, Po) < 1e-6 and rel_err(Bi, Bo) < 1e-6 and rel_err(Q, Qo) < 1e-6 and abs(half - half_o) < 1e-6 * half_o
    assert abs(sp - (P * P).sum()) < 1e-10 * sp and abs(sb - Bi.dot(Bi)) < 1e-10 * sb
    if round_users == 1 and iters == 1:
        assert rel_err(P, z['P']) < 1e-6 and rel_err(Q, z['Q']) < 1e-6         # one user per round: the reference's pass


def test_rounds_on_ragged_users_with_repeats(dev):
    # users without / with one event, duplicates inside a user, negatives that repeat, k = 200 (four registers per lane and row)
    from oracle.numpy_fism import fism_rounds
    rng = np.random.RandomState(7)
    n, k, rho, alpha = 60, 200, 3, 0.6
    sizes = [0, 1, 5, 12, 1, 30, 2, 0, 7, 9, 3]
    ptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    ev_i = np.concatenate([rng.randint(0, n // 2, s) for s in sizes]).astype(np.int32)
    negs = []
    for u, s in enumerate(sizes):
        if s > 1:
            mine = set(ev_i[ptr[u]:ptr[u + 1]].tolist())
            negs += [int(rng.choice([x for x in range(n) if x not in mine])) for _ in range(s * rho)]
    negs = np.array(negs, np.int32)
    P0, Q0, B0 = rng.rand(n, k) / 100, (rng.rand(n, k) / 10).astype(np.float32), rng.rand(n) / 100
    for round_users in (1, 4, 100):
        dev.fism_set_model(P0, Q0, B0)
        half, _, _, _ = dev.fism_rounds(ptr, ev_i, negs, rho, coefs(ptr, alpha), round_users, 0.02, 0.01, 0.03)
        Po, Qo, Bo = P0.copy(), Q0.copy(), B0.copy()
        half_o = fism_rounds(Po, Qo, Bo, ptr, ev_i, negs, rho, alpha, 0.02, 0.01, 0.03, round_users)
        P, Q, Bi = np.empty_like(P0), np.empty_like(Q0), np.empty_like(B0)
        dev.fism_get_model(P, Q, Bi)
        assert rel_err(P, Po) < 1e-6 and rel_err(Bi, Bo) < 1e-6 and rel_err(Q, Qo) < 1e-6 and abs(half - half_o) < 1e-6 * half_o, round_users


def test_through_the_plugin_surface(tmp_path, capsys):
    # FISM.conf keys on the C1 log, driven as tools/make_goldens.py drives the reference's class
    from yue_amd import synth
    from yue_amd.recommender.cf.FISM import FISM
    from yue_amd.tool.config import Config
    z, meta = gz('g8_fism_c1_k10_e2.npz'), gj('g8_fism_c1_k10_e2.json')
    log = tmp_path / 'log.txt'
    synth.write_text_log(str(log), 1000, 1000, 20)
    text = _conf_text({'record': str(log), 'recommender': 'FISM', 'num.factors': '10', 'num.max.iter': '2', 'item.ranking': '-topN 5,10',
                       'learnRate': '-init 0.015 -max 1', 'reg.lambda': '-u 0.01 -i 0.01 -b 0.01 -s 0.01',
                       'output.setup': 'on -dir ' + str(tmp_path / 'results') + '/'}, {'FISM': '-rho 2 -alpha 0.5'})
    path = tmp_path / 'fism.conf'
    path.write_text(text)
    conf = Config(str(path))
    rec = FISM(conf, _load(conf), [])
    rec.readConfiguration()
    random.seed(int(z['seed']))
    np.random.seed(int(z['seed']))
    rec.initModel()
    assert np.array_equal(rec.P, z['P0']) and np.array_equal(rec.Q, z['Q0']) and np.array_equal(rec.Bi, z['B0'])
    capsys.readouterr()
    rec.buildModel()
    lines = [ln for ln in capsys.readouterr().out.splitlines() if 'iteration' in ln]
    assert lines == meta['lines']
    assert rel_err(rec.P, z['P']) < 1e-11 and rel_err(rec.Bi, z['Bi']) < 1e-11 and rel_err(rec.Q, z['Q']) < 1e-6
    assert rec.lRate == float(z['lRate'])
    rec.evalRanking()
    assert rec.measure == meta['measure']
    assert glob.glob(str(tmp_path / 'results' / 'FISM@*items*.txt'))
    users = list(rec.data.testSet.keys())
    s = rec.predict(users[3])
    assert s.dtype == np.float64 and np.abs(s - z['predict16'][3]).max() < 1e-12


@pytest.mark.parametrize('k', [64, 130, 10])
def test_rounds_with_working_rows_in_lds_equal_the_global_form(dev, k):
    # users of at most 64 touches take k_fism_round_lds (the wave finds its rows itself, working copies in LDS): same pass as
    # k_fism_round (option fism_lds = 0) and as the NumPy oracle; ragged users, duplicates inside a user, repeated negatives;
    # with and without the rows only one user of a round touches stored in place (option fism_inplace: one user per round =
    # every row in place, 5 = some of them, 100 = the users of the whole problem share most rows).  k = 64 and 10: the round-start
    # rows stay in LDS beside the working rows (k_fism_round_lds<KR, true>); k = 130: 100 KB of working rows, twice that does not
    # fit in 160 KB -> <4, false>, the model rows are read again for the differences
    from oracle.numpy_fism import fism_rounds
    rng = np.random.RandomState(11 + k)
    n, rho, alpha = 50, 3, 0.5
    sizes = [0, 1, 5, 12, 1, 16, 2, 0, 7, 9, 3, 16, 16]
    ptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    ev_i = np.concatenate([rng.randint(0, n // 2, s) for s in sizes]).astype(np.int32)
    negs = []
    for u, s in enumerate(sizes):
        if s > 1:
            mine = set(ev_i[ptr[u]:ptr[u + 1]].tolist())
            negs += [int(rng.choice([x for x in range(n) if x not in mine])) for _ in range(s * rho)]
    negs = np.array(negs, np.int32)
    P0, Q0, B0 = rng.rand(n, k) / 100, (rng.rand(n, k) / 10).astype(np.float32), rng.rand(n) / 100
    for round_users in (1, 5, 100):
        res = []
        for lds, inplace in ((1, 1), (1, 0), (0, 1)):
            dev.set_option('fism_lds', lds)
            dev.set_option('fism_inplace', inplace)
            dev.fism_set_model(P0, Q0, B0)
            half, _, _, _ = dev.fism_rounds(ptr, ev_i, negs, rho, coefs(ptr, alpha), round_users, 0.02, 0.01, 0.03)
            P, Q, Bi = np.empty_like(P0), np.empty_like(Q0), np.empty_like(B0)
            dev.fism_get_model(P, Q, Bi)
            res.append((half, P, Q, Bi))
        dev.set_option('fism_lds', 1)
        dev.set_option('fism_inplace', 1)
        Po, Qo, Bo = P0.copy(), Q0.copy(), B0.copy()
        half_o = fism_rounds(Po, Qo, Bo, ptr, ev_i, negs, rho, alpha, 0.02, 0.01, 0.03, round_users)
        for half, P, Q, Bi in res:
            assert rel_err(P, Po) < 1e-6 and rel_err(Bi, Bo) < 1e-6 and rel_err(Q, Qo) < 1e-6 and abs(half - half_o) < 1e-6 * half_o, round_users


@pytest.mark.parametrize('k,round_users', [(64, 96), (32, 250), (130, 40)])
def test_many_rounds_with_popular_items_match_the_numpy_oracle(dev, k, round_users):
    # 1,500 ragged users on 300 items with a popular head over many rounds (a last round shorter than the others): rows in
    # place, contended differences and the count arrays that swap from round to round, against the NumPy oracle of the round form;
    # and the forms of the device path (rows in place or not, working rows in LDS or global memory) against each other
    from oracle.numpy_fism import fism_rounds
    rng = np.random.RandomState(500 + k)
    m, n, rho, alpha = 1500, 300, 2, 0.5
    sizes = rng.randint(0, 13, m)
    ptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    pop = 1.0 / np.arange(1, n + 1) ** 0.8
    pop /= pop.sum()
    ev_i = rng.choice(n, size=int(ptr[-1]), p=pop).astype(np.int32)
    negs = []
    for u in range(m):
        if sizes[u] > 1:
            mine = ev_i[ptr[u]:ptr[u + 1]]
            d = rng.randint(0, n, sizes[u] * rho)
            while True:
                bad = np.isin(d, mine)
                if not bad.any():
                    break
                d[bad] = rng.randint(0, n, int(bad.sum()))
            negs.append(d)
    negs = np.concatenate(negs).astype(np.int32)
    P0, Q0, B0 = rng.rand(n, k) / 100, (rng.rand(n, k) / 10).astype(np.float32), rng.rand(n) / 100
    lr, reg_i, reg_b = 0.002, 0.01, 0.01
    Po, Qo, Bo = P0.copy(), Q0.copy(), B0.copy()
    half_o = fism_rounds(Po, Qo, Bo, ptr, ev_i, negs, rho, alpha, lr, reg_i, reg_b, round_users)
    assert np.isfinite(half_o)
    try:
        for lds, inplace in ((1, 1), (1, 0), (0, 1)):
            dev.set_option('fism_lds', lds)
            dev.set_option('fism_inplace', inplace)
            dev.fism_set_model(P0, Q0, B0)
            half, _, _, _ = dev.fism_rounds(ptr, ev_i, negs, rho, coefs(ptr, alpha), round_users, lr, reg_i, reg_b)
            P, Q, Bi = np.empty_like(P0), np.empty_like(Q0), np.empty_like(B0)
            dev.fism_get_model(P, Q, Bi)
            assert rel_err(P, Po) < 1e-6 and rel_err(Bi, Bo) < 1e-6 and rel_err(Q, Qo) < 1e-6 and abs(half - half_o) < 1e-6 * half_o, (lds, inplace)
    finally:
        dev.set_option('fism_lds', 1)
        dev.set_option('fism_inplace', 1)
