"""GPU parity of predict() and the evalRanking selection, through the C ABI.

Integer ranking lists must be bit-exact (BASELINE.json north_star): the GPU scores use the same
k-ascending fp32 fma chain as the oracle, so lists AND scores are compared for equality.  On the
reference's own trained factors the lists are also compared with the reference's golden lists.
"""
import numpy as np
import pytest

from util import csr_from_events, gz, mask_rows

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    from yue_amd._shim import Device
    d = Device(0, raise_errors=True)
    yield d
    d.close()


def _rand_problem(m, n, k, per_user, seed, signed=True):
    rs = np.random.RandomState(seed)
    P = rs.randn(m, k).astype(np.float32) if signed else (rs.rand(m, k).astype(np.float32) / 10)
    Q = rs.randn(n, k).astype(np.float32) if signed else (rs.rand(n, k).astype(np.float32) / 10)
    rows = [np.sort(rs.choice(n, size=min(n - 1, rs.randint(0, per_user + 1)), replace=False)) for _ in range(m)]
    indptr = np.cumsum([0] + [len(r) for r in rows]).astype(np.int64)
    indices = (np.concatenate(rows) if indptr[-1] else np.zeros(0)).astype(np.int32)
    return P, Q, indptr, indices


def test_predict_is_bit_exact_with_oracle(dev, orc):
    for (m, n, k) in [(5, 1000, 10), (3, 777, 64), (4, 2049, 128), (2, 50, 33)]:
        P, Q, _, _ = _rand_problem(m, n, k, 0, seed=k)
        dev.set_factors(P, Q)
        for u in range(m):
            assert np.array_equal(dev.scores(u), orc.scores(P, Q, u))


@pytest.mark.parametrize('tag,N', [('c1_top10', 10), ('c1_top20', 20)])
def test_lists_match_reference_golden(dev, orc, tag, N):
    # evalRanking (IterativeRecommender.py:93-145) on the reference's trained factors
    z = gz('g4_c1_k10_e1.npz')
    ev = gz('g2_events_c1.npz')
    g = gz('g5_%s.npz' % tag)
    m = int(z['m'])
    indptr, indices = csr_from_events(ev['ev_u'], ev['ev_i'], m)
    ev_ptr = np.zeros(m + 1, np.int64)
    np.add.at(ev_ptr, ev['ev_u'] + 1, 1)
    ev_ptr = np.cumsum(ev_ptr)
    dev.set_factors(z['P'], z['Q'])
    dev.set_interactions(indptr, indices, ev_ptr, ev['ev_i'])
    users = g['test_users']
    ids, sc = dev.topn_scan(users, N)                       # mask = uploaded training items
    assert np.array_equal(ids, g['rec_ids'])                # the reference's integer lists
    mp, mi = mask_rows(indptr, indices, users)
    oid, osc, rc = orc.topn_scan(z['P'], z['Q'], users, N, mp, mi)
    assert rc == 0 and np.array_equal(ids, oid) and np.array_equal(sc, osc)
    ids2, sc2 = dev.topn_scan(users, N, mp, mi)             # explicit mask, same rows
    assert np.array_equal(ids2, ids) and np.array_equal(sc2, sc)


@pytest.mark.parametrize('m,n,k,N,per_user', [
    (130, 1000, 10, 10, 30),      # two workgroups, partial second one
    (37, 333, 64, 5, 300),        # n not a multiple of 32, dense masks
    (300, 4099, 128, 20, 60),
    (64, 2000, 128, 100, 10),     # largest N
    (33, 500, 16, 1, 5),
    (10, 97, 7, 3, 20),           # odd k (zero padded), tiny
    (70, 1500, 32, 10, 40),
    (40, 900, 48, 10, 40),        # k not in the bf16 set -> f32 kernel
    (50, 1200, 200, 10, 30),      # k > 128 (trainable up to 256): f32 kernel with 128 k-steps
    (20, 700, 256, 20, 30),
])
def test_scan_matches_oracle(dev, orc, m, n, k, N, per_user):
    P, Q, indptr, indices = _rand_problem(m, n, k, per_user, seed=m + n + k + N)
    dev.set_factors(P, Q)
    users = np.random.RandomState(3).permutation(m).astype(np.int32)
    mp, mi = mask_rows(indptr, indices, users)
    ids, sc = dev.topn_scan(users, N, mp, mi)
    oid, osc, rc = orc.topn_scan(P, Q, users, N, mp, mi)
    assert rc == 0
    assert np.array_equal(ids, oid)
    assert np.array_equal(sc, osc)
    ms, events, rescored, used_bf16 = dev.scan_stats()
    assert ms > 0 and events >= len(users) * N
    assert used_bf16 == (k in (16, 32, 64, 128)) and (not used_bf16 or rescored >= events)
    # both bf16 pre-filter kernels (two tiles per iteration / one) give the same lists and scores
    dev.set_option('scan_batch', 1)
    ids1, sc1 = dev.topn_scan(users, N, mp, mi)
    dev.set_option('scan_batch', 0)
    assert np.array_equal(ids1, ids) and np.array_equal(sc1, sc)
    # the exact f32-MFMA kernel and the bf16 pre-filter path give the same lists and scores
    dev.set_option('scan_f32', 1)
    ids2, sc2 = dev.topn_scan(users, N, mp, mi)
    dev.set_option('scan_f32', 0)
    assert not dev.scan_stats()[3]
    assert np.array_equal(ids2, ids) and np.array_equal(sc2, sc)


@pytest.mark.parametrize('case', range(12))
def test_scan_on_random_shapes(dev, orc, case):
    # random shapes through both bf16 kernels and the f32 kernel: users not a multiple of the workgroup, item counts around
    # the tile / stage boundaries (32, 64), short lists, dense masks, positive and signed factors, trained-like norms
    rs = np.random.RandomState(500 + case)
    k = int(rs.choice([16, 32, 64, 128, 128, 24, 200]))
    N = int(rs.randint(1, 41))
    n = int(rs.choice([N + 33, 64 * rs.randint(2, 40), 64 * rs.randint(2, 40) + rs.randint(1, 64), rs.randint(N + 40, 6000)]))
    m = int(rs.randint(1, 700))
    per_user = int(rs.randint(0, min(200, n - N - 1)))
    P, Q, indptr, indices = _rand_problem(m, n, k, per_user, seed=900 + case, signed=bool(case % 2))
    if case % 3 == 0:
        Q *= rs.rand(n, 1).astype(np.float32) * 3            # uneven item norms: the norm-bound skips and exits take effect
    dev.set_factors(P, Q)
    users = rs.permutation(m)[:max(1, int(rs.randint(1, m + 1)))].astype(np.int32)
    mp, mi = mask_rows(indptr, indices, users)
    oid, osc, rc = orc.topn_scan(P, Q, users, N, mp, mi)
    assert rc == 0
    for batch in (0, 1):
        dev.set_option('scan_batch', batch)
        ids, sc = dev.topn_scan(users, N, mp, mi)
        assert np.array_equal(ids, oid) and np.array_equal(sc, osc), (case, k, N, n, m, batch)
    dev.set_option('scan_batch', 0)
    dev.set_option('scan_f32', 1)
    ids, sc = dev.topn_scan(users, N, mp, mi)
    dev.set_option('scan_f32', 0)
    assert np.array_equal(ids, oid) and np.array_equal(sc, osc), (case, k, N, n, m, 'f32')


def test_true_topn_mode(dev, orc):
    # a real top-N beside the reference's overwrite-scan (which is not one, F4)
    for (m, n, k, N, quant) in [(100, 3000, 128, 20, False), (50, 700, 10, 10, False), (40, 600, 16, 10, True)]:
        P, Q, indptr, indices = _rand_problem(m, n, k, 25, seed=n + N)
        if quant:       # many equal scores: ties must keep the lower item id first
            P, Q = np.round(P), np.round(Q)
        dev.set_factors(P, Q)
        users = np.arange(m, dtype=np.int32)
        mp, mi = mask_rows(indptr, indices, users)
        dev.set_option('topn_true', 1)
        ids, sc = dev.topn_scan(users, N, mp, mi)
        dev.set_option('topn_true', 0)
        oid, osc, _ = orc.topn_true(P, Q, users, N, mp, mi)
        assert np.array_equal(ids, oid) and np.array_equal(sc, osc)
        for t in range(0, m, 7):                                # independent check with NumPy
            s = orc.scores(P, Q, t).astype(np.float64)
            s[mi[mp[t]:mp[t + 1]]] = -np.inf
            order = np.lexsort((np.arange(n), -s))[:N]
            assert np.array_equal(ids[t], order)


def test_scan_with_ties_and_duplicates(dev, orc):
    # quantised factors give many equal scores: ties go after equals, stable seed order
    rs = np.random.RandomState(9)
    m, n, k, N = 40, 600, 8, 10
    P = rs.randint(-2, 3, size=(m, k)).astype(np.float32)
    Q = rs.randint(-2, 3, size=(n, k)).astype(np.float32)
    indptr = np.zeros(m + 1, np.int64)
    indices = np.zeros(0, np.int32)
    dev.set_factors(P, Q)
    users = np.arange(m, dtype=np.int32)
    ids, sc = dev.topn_scan(users, N, indptr, indices)
    oid, osc, _ = orc.topn_scan(P, Q, users, N, indptr, indices)
    assert np.array_equal(ids, oid) and np.array_equal(sc, osc)


def test_too_few_candidates_raises_like_the_reference(dev):
    P, Q, _, _ = _rand_problem(2, 12, 4, 0, seed=1)
    dev.set_factors(P, Q)
    mp = np.array([0, 9, 9], np.int64)
    mi = np.arange(9, dtype=np.int32)
    with pytest.raises(IndexError):                      # IterativeRecommender.py:126
        dev.topn_scan(np.array([0, 1], np.int32), 5, mp, mi)
    ids, _ = dev.topn_scan(np.array([1], np.int32), 5, np.array([0, 0], np.int64), np.zeros(0, np.int32))
    assert ids.shape == (1, 5) and (ids >= 0).all()


def test_full_size_scan_properties(dev, orc):
    # C5-like slice: 200K items, k=128, N=20; properties + oracle on a few users
    rs = np.random.RandomState(5)
    m, n, k, N = 512, 200000, 128, 20
    P = rs.rand(m, k).astype(np.float32) / 10
    Q = rs.rand(n, k).astype(np.float32) / 10
    indptr = np.arange(m + 1, dtype=np.int64) * 40
    indices = np.sort(rs.randint(0, n, size=(m, 40)), axis=1)
    indices[:, 1:] += (np.diff(indices, axis=1) == 0)          # crude de-dup keeps rows strictly sorted mostly
    indices = np.sort(indices, axis=1)
    ok = (np.diff(indices, axis=1) > 0).all(axis=1)
    users = np.where(ok)[0].astype(np.int32)
    dev.set_factors(P, Q)
    mp, mi = mask_rows(indptr, indices.reshape(-1).astype(np.int32), users)
    ids, sc = dev.topn_scan(users, N, mp, mi)
    dev.set_option('scan_batch', 1)
    ids1, sc1 = dev.topn_scan(users, N, mp, mi)
    dev.set_option('scan_batch', 0)
    assert np.array_equal(ids1, ids) and np.array_equal(sc1, sc)
    # slot 0 is the global maximum over candidates; scores are non-increasing; no masked item listed
    assert (np.diff(sc, axis=1) <= 0).all()
    for t in range(0, len(users), 37):
        s = dev.scores(int(users[t]))
        s[mi[mp[t]:mp[t + 1]]] = -np.inf
        assert ids[t, 0] == int(np.argmax(s)) and sc[t, 0] == s.max()
        assert not set(ids[t].tolist()) & set(mi[mp[t]:mp[t + 1]].tolist())
        assert np.array_equal(sc[t], dev.scores(int(users[t]))[ids[t]])
    sub = slice(0, 24)
    oid, osc, _ = orc.topn_scan(P, Q, users[sub], N, *mask_rows(indptr, indices.reshape(-1).astype(np.int32), users[sub]))
    assert np.array_equal(ids[sub], oid) and np.array_equal(sc[sub], osc)


def test_config5_at_its_stated_size(orc):
    """BASELINE config 5 in full: 1M users x 200K items, k = 128, top-20, training items masked (the synthetic interactions and
    factors of bench.py --workload c5).  Size-independent properties for ALL users -- ids in range, scores non-increasing, no
    training item listed -- and, for sampled users, slot 0 = the arg-max of predict() over the candidates, listed scores =
    predict()[ids] and the whole list equal to the oracle's."""
    from yue_amd import synth
    from yue_amd._shim import Device
    m, n, d, k, N = 1000000, 200000, 50, 128, 20
    data = synth.make_arrays(m, n, d, seed=20260001)
    P, Q = synth.init_factors(m, n, k, 20260002)
    dev = Device(0, raise_errors=True)
    try:
        dev.set_factors(P, Q)
        dev.set_interactions(data['indptr'], data['indices'], data['ev_ptr'], data['ev_i'])
        users = np.arange(m, dtype=np.int32)
        ids, sc = dev.topn_scan(users, N)
        ms, events, rescored, used_bf16 = dev.scan_stats()
        print('full config 5: %.1f ms in the kernel, %.1f state-machine events and %.1f exact re-scores per user' % (ms, events / m, rescored / m))
        assert used_bf16 and events > 0 and rescored >= events - m * N
        assert ids.min() >= 0 and ids.max() < n
        assert (np.diff(sc, axis=1) <= 0).all()
        # no training item in any list: (user, item) keys of the lists against the sorted keys of the CSR
        train_keys = np.repeat(np.arange(m, dtype=np.int64), np.diff(data['indptr'])) * n + data['indices']
        list_keys = (np.arange(m, dtype=np.int64)[:, None] * n + ids).ravel()
        pos = np.searchsorted(train_keys, list_keys)
        pos[pos == len(train_keys)] = 0
        assert not (train_keys[pos] == list_keys).any()
        sample = np.linspace(0, m - 1, 32).astype(np.int32)
        for u in sample:
            s = dev.scores(int(u))
            s[data['indices'][data['indptr'][u]:data['indptr'][u + 1]]] = -np.inf
            assert ids[u, 0] == int(np.argmax(s)) and sc[u, 0] == s.max()
            assert np.array_equal(sc[u], dev.scores(int(u))[ids[u]])
        oid, osc, rc = orc.topn_scan(P, Q, sample, N, *mask_rows(data['indptr'], data['indices'], sample))
        assert rc == 0 and np.array_equal(ids[sample], oid) and np.array_equal(sc[sample], osc)
    finally:
        dev.close()


@pytest.mark.parametrize('m,n,k,N,per_user,signed', [(300, 40000, 128, 20, 60, True), (70, 33000, 64, 5, 2000, False), (130, 20000, 16, 64, 30, True),
                                                   (40, 50001, 32, 10, 0, False), (257, 17000, 128, 1, 100, True)])
@pytest.mark.parametrize('form', [3, 2, 1])
def test_two_phase_scan_equals_the_fused_kernel_and_the_oracle(dev, orc, m, n, k, N, per_user, signed, form):
    """Catalogues of 16,384 items and more take the chunked path (k_topn_scan_bf16p for the first 512 items, then
    k_scan_filter + k_scan_select per chunk of doubling size): lists and scores must equal the oracle's and the fused
    kernel's, with the overwrite-scan and as a true top-N, with ties, heavy masks (more than fit in LDS) and odd sizes; in
    all forms of the filter (scan_filter_ub = 3: two blocks of 32 users per wave and item rows by DMA, the default; 2: rows
    through registers; 1: one block, eight waves)."""
    P, Q, indptr, indices = _rand_problem(m, n, k, per_user, seed=1000 + k + N, signed=signed)
    Q[5000:5003] = Q[123]                                      # equal rows: ties between far-apart items
    users = np.arange(m, dtype=np.int32)[::-1].copy()
    mp, mi = mask_rows(indptr, indices, users)
    dev.set_factors(P, Q)
    for true_topn in (0, 1):
        dev.set_option('topn_true', true_topn)
        dev.set_option('scan_filter_ub', form)
        try:
            ids, sc = dev.topn_scan(users, N, mp, mi)
            assert dev.get_option('scan_last_chunks') >= 2
            ms, events, rescored, used_bf16 = dev.scan_stats()
            dev.set_option('scan_two_phase', 0)
            ids_f, sc_f = dev.topn_scan(users, N, mp, mi)
            assert dev.get_option('scan_last_chunks') == 0
            _, events_f, _, _ = dev.scan_stats()
        finally:
            dev.set_option('scan_two_phase', 1)
            dev.set_option('topn_true', 0)
            dev.set_option('scan_filter_ub', 3)
        assert np.array_equal(ids, ids_f) and np.array_equal(sc, sc_f) and events == events_f
        oid, osc, rc = (orc.topn_true if true_topn else orc.topn_scan)(P, Q, users, N, mp, mi)
        assert np.array_equal(ids, oid) and np.array_equal(sc, osc)


@pytest.mark.parametrize('form', [3, 2, 1])
@pytest.mark.parametrize('m,n,k,N', [(1500, 60000, 128, 20), (700, 40000, 64, 10)])
def test_two_phase_scan_with_settled_users(dev, orc, m, n, k, N, form):
    """Item norms that fall along the catalogue (what a few epochs of training leave on popularity-ordered items): most users
    are settled against the tail after the first chunks, the filter runs in its variant whose workgroups stop then
    (k_scan_filter<.., SETTLE>); some users get a late strong item so that not every workgroup stops at the same place."""
    rs = np.random.RandomState(5 + k)
    P = (rs.rand(m, k).astype(np.float32) - 0.3) / 4
    Q = (rs.rand(n, k).astype(np.float32) - 0.3) / 4
    Q *= (1.0 / (1.0 + np.arange(n, dtype=np.float32) / 300.0))[:, None]
    P[3 * m // 4:] -= P[3 * m // 4:].mean(axis=1, keepdims=True)     # a quarter of the users with low scores for their norm: they settle late or never
    late = ((3, 3000), (700 % m, 4000), (m - 1, 3500))
    for u, it in late:
        Q[it] = 0.5 * P[u] / np.linalg.norm(P[u])               # an item of the second chunk that must still enter user u's list
    indptr = np.arange(m + 1, dtype=np.int64) * 4
    indices = (rs.randint(0, n // 4, size=(m, 4)) + np.arange(4) * (n // 4)).astype(np.int32).reshape(-1)     # sorted, distinct per row
    indices[0:4] = [0, 1, 2, 3]
    users = np.arange(m, dtype=np.int32)
    mp, mi = mask_rows(indptr, indices, users)
    dev.set_factors(P, Q)
    dev.set_option('scan_filter_ub', form)
    try:
        ids, sc = dev.topn_scan(users, N, mp, mi)
    finally:
        dev.set_option('scan_filter_ub', 3)
    assert dev.get_option('scan_last_chunks') >= 2 and dev.get_option('scan_last_settle') == 1
    done, total = dev.scan_work()
    assert done < 0.5 * total                                   # most tiles were never scored
    oid, osc, rc = orc.topn_scan(P, Q, users, N, mp, mi)
    assert np.array_equal(ids, oid) and np.array_equal(sc, osc)
    assert all(it in ids[u] for u, it in late)
    dev.set_option('scan_two_phase', 0)
    try:
        ids_f, sc_f = dev.topn_scan(users, N, mp, mi)
    finally:
        dev.set_option('scan_two_phase', 1)
    assert np.array_equal(ids, ids_f) and np.array_equal(sc, sc_f)


def test_two_phase_scan_with_slabs_on_two_streams(dev, orc):
    """Calls with many users split them into four or more slabs that alternate between two streams (a slab's selection beside the
    next slab's filter, each stream with its own survivor words).  Here the user count that switches this on is lowered: 3,000
    users = 4 slabs of 768 (the last one short); lists and scores must equal the one-stream run and the oracle."""
    m, n, k, N = 3000, 20000, 128, 20
    P, Q, indptr, indices = _rand_problem(m, n, k, 25, seed=77, signed=True)
    users = np.arange(m, dtype=np.int32)
    mp, mi = mask_rows(indptr, indices, users)
    dev.set_factors(P, Q)
    dev.set_option('scan_streams_min_users', 1024)
    try:
        ids, sc = dev.topn_scan(users, N, mp, mi)
        assert dev.get_option('scan_last_chunks') >= 2
        dev.set_option('scan_streams', 1)
        ids1, sc1 = dev.topn_scan(users, N, mp, mi)
    finally:
        dev.set_option('scan_streams', 2)
        dev.set_option('scan_streams_min_users', 262144)
    assert np.array_equal(ids, ids1) and np.array_equal(sc, sc1)
    sample = np.arange(0, m, 7, dtype=np.int32)
    smp, smi = mask_rows(indptr, indices, sample)
    oid, osc, rc = orc.topn_scan(P, Q, sample, N, smp, smi)
    assert np.array_equal(ids[sample], oid) and np.array_equal(sc[sample], osc)


def test_two_phase_scan_with_a_heavy_listener_of_the_catalogues_head(dev, orc):
    """A few users have listened to almost all of the first 512 items: their first chunk holds fewer than N candidates.  Only they
    go through the fused kernel over all items (option-free: scan_last_few_users says how many), everybody else through the
    filter / select pair; every list and score equals the oracle's.  A user with too few candidates in the WHOLE catalogue still
    raises, as the reference does (base/IterativeRecommender.py:126)."""
    m, n, k, N = 6000, 20000, 64, 20
    P, Q, indptr, indices = _rand_problem(m, n, k, 30, seed=77, signed=True)
    # users 5, 4000 and 5999: the first 600 items minus a handful, plus their own
    rows = [indices[indptr[u]:indptr[u + 1]] for u in range(m)]
    rs = np.random.RandomState(3)
    for u, keep in ((5, 7), (4000, 19), (5999, 0)):
        head = np.setdiff1d(np.arange(600, dtype=np.int32), rs.choice(600, size=keep, replace=False).astype(np.int32))
        rows[u] = np.union1d(rows[u], head).astype(np.int32)
    indptr2 = np.zeros(m + 1, np.int64)
    indptr2[1:] = np.cumsum([len(r) for r in rows])
    indices2 = np.concatenate(rows).astype(np.int32)
    dev.set_factors(P, Q)
    dev.set_interactions(indptr2, indices2, indptr2, indices2)       # (events = the listened items, once each: only the mask matters here)
    users = np.arange(m, dtype=np.int32)
    ids, sc = dev.topn_scan(users, N)
    assert dev.get_option('scan_last_chunks') > 0 and dev.get_option('scan_last_few_users') == 3
    mp, mi = mask_rows(indptr2, indices2, users)
    oid, osc, rc = orc.topn_scan(P, Q, users, N, mp, mi)
    assert rc == 0 and np.array_equal(ids, oid) and np.array_equal(sc, osc)
    # one user who has listened to everything but N - 1 items
    rows[17] = np.setdiff1d(np.arange(n, dtype=np.int32), np.arange(100, 100 + N - 1, dtype=np.int32))
    indptr3 = np.zeros(m + 1, np.int64)
    indptr3[1:] = np.cumsum([len(r) for r in rows])
    indices3 = np.concatenate(rows).astype(np.int32)
    dev.set_interactions(indptr3, indices3, indptr3, indices3)
    with pytest.raises(IndexError):
        dev.topn_scan(users, N)
