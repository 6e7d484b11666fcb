"""GPU parity of the EXACT training path (chain_kernels.hpp: the reference's sequential loop, recommender/cf/BPR.py:40-62,
as a dataflow launch), through the C ABI.  Checker: oracle/bpr_oracle.c: orc_bpr_sequential (pinned to the reference by
tests/test_oracle_golden.py) and the reference's own golden factors.  Same dot order on both sides: differences can only come
from exp / log last-bit rounding, so the bound is 1e-6 (north_star: 1e-5 rel)."""
import numpy as np
import pytest

from yue_amd import synth
from util import gz, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    from yue_amd._shim import Device
    d = Device(0, raise_errors=True)
    yield d
    d.close()


def _problem(m, n, d, k, seed):
    data = synth.make_arrays(m, n, d, seed=seed)
    ev_u = np.repeat(np.arange(m, dtype=np.int32), np.diff(data['ev_ptr']))
    P0, Q0 = synth.init_factors(m, n, k, seed + 1)
    return data, ev_u, P0, Q0


@pytest.mark.parametrize('m,n,d,k', [(3000, 2000, 20, 128), (5000, 64, 12, 64), (700, 900, 30, 10), (400, 300, 25, 200), (20000, 5000, 20, 128)])
def test_exact_epoch_equals_the_sequential_loop(dev, orc, m, n, d, k):
    """yue_bpr_epoch with option epoch_exact: the device sampler's negatives, applied strictly in the reference's order.
    (5000 users on 64 items: every row is hot, the whole epoch is a handful of long chains.)"""
    data, ev_u, P0, Q0 = _problem(m, n, d, k, 31 + k)
    dev.set_factors(P0, Q0)
    dev.set_interactions(data['indptr'], data['indices'], data['ev_ptr'], data['ev_i'])
    dev.set_option('epoch_exact', 1)
    try:
        Po, Qo = P0.copy(), Q0.copy()
        for ep in range(2):
            j = dev.sample_negatives(5, ep)
            nll, sp, sq = dev.bpr_epoch(5, ep, 0, 0.02, 0.01, 0.01)
            nll_o = orc.bpr_sequential(Po, Qo, ev_u, data['ev_i'], j, 0.02, 0.01, 0.01)
            assert abs(nll - nll_o) <= 1e-9 * abs(nll_o)
        P, Q = dev.get_factors()
        assert rel_err(P, Po) < 1e-6 and rel_err(Q, Qo) < 1e-6
        assert abs(sp - orc.sumsq(P)) <= 1e-12 * sp and abs(sq - orc.sumsq(Q)) <= 1e-12 * sq
        print('bit-equal to the oracle: P %.4f Q %.4f; runs %d on %d waves' % (np.mean(P == Po), np.mean(Q == Qo), dev.get_option('chain_last_runs'), dev.get_option('chain_last_waves')))
    finally:
        dev.set_option('epoch_exact', 0)


def test_replay_of_an_interleaved_stream(dev, orc):
    """yue_bpr_replay on a stream that is NOT user-major: users come back in later runs (user rows versioned per run), items
    repeat inside a run, some triplets are skipped (j < 0)."""
    rs = np.random.RandomState(77)
    m, n, k, T = 300, 500, 64, 40000
    P0, Q0 = synth.init_factors(m, n, k, 78)
    u = rs.randint(0, m, size=T).astype(np.int32)
    u[1000:1400] = 7                                       # one long run
    idx = np.arange(0, T - 1, 3)
    u[idx] = u[idx + 1]                                    # many runs of two
    i = (n * rs.rand(T) ** 2).astype(np.int32)              # skewed positives
    j = rs.randint(0, n, size=T).astype(np.int32)
    j[j == i] = (i[j == i] + 1) % n
    j[rs.rand(T) < 0.01] = -1
    dev.set_factors(P0, Q0)
    nll = dev.bpr_replay(u, i, j, 0.03, 0.02, 0.01)
    Po, Qo = P0.copy(), Q0.copy()
    nll_o = orc.bpr_sequential(Po, Qo, u, i, j, 0.03, 0.02, 0.01)
    P, Q = dev.get_factors()
    assert abs(nll - nll_o) <= 1e-9 * abs(nll_o)
    assert rel_err(P, Po) < 1e-6 and rel_err(Q, Qo) < 1e-6
    # the levelled replay of round 1 (one launch per dependency level) lands on the same factors
    dev.set_factors(P0, Q0)
    dev.set_option('replay_levels', 1)
    try:
        nll_l = dev.bpr_replay(u, i, j, 0.03, 0.02, 0.01)
        levels = dev.get_option('replay_last_levels')
    finally:
        dev.set_option('replay_levels', 0)
    Pl, Ql = dev.get_factors()
    assert np.array_equal(P, Pl) and np.array_equal(Q, Ql) and abs(nll_l - nll) <= 1e-9 * abs(nll)
    print('dependency levels of the stream: %d' % levels)


def test_exact_path_on_the_reference_goldens(dev):
    """Two epochs of the reference's own run (k = 128): the factors the reference's NumPy loop produced, within 1e-5."""
    z = gz('g4_d3_k128_e2.npz')
    iters = int(z['iters'])
    E = len(z['u']) // iters
    P0, Q0 = synth.init_factors(int(z['m']), int(z['n']), int(z['k']), int(z['seed']))
    dev.set_factors(P0, Q0)
    for ep in range(iters):
        sl = slice(ep * E, (ep + 1) * E)
        dev.bpr_replay(z['u'][sl], z['i'][sl], z['j'][sl], 0.02, 0.01, 0.01)
    P, Q = dev.get_factors()
    assert rel_err(P, z['P']) < 1e-5 and rel_err(Q, z['Q']) < 1e-5


def test_exact_path_rejects_bad_streams_before_touching_the_factors(dev):
    from yue_amd._shim import YueHipError
    P0, Q0 = synth.init_factors(50, 40, 16, 3)
    dev.set_factors(P0, Q0)
    good = (np.array([1, 1, 2], np.int32), np.array([3, 4, 5], np.int32), np.array([6, 7, 8], np.int32))
    for bad in [(np.array([1, 50, 2], np.int32), good[1], good[2]),             # user out of range
                (good[0], np.array([3, 40, 5], np.int32), good[2]),             # item out of range
                (good[0], good[1], np.array([6, 4, 8], np.int32)),              # i == j
                (good[0], good[1], np.array([6, 7, 41], np.int32))]:
        with pytest.raises(YueHipError):
            dev.bpr_replay(bad[0], bad[1], bad[2], 0.05, 0.02, 0.03)
        P, Q = dev.get_factors()
        assert np.array_equal(P, P0) and np.array_equal(Q, Q0)
    assert dev.bpr_replay(*good, 0.05, 0.02, 0.03) > 0.0


def test_a_wave_that_cannot_get_its_row_gives_up_instead_of_hanging(dev):
    """The guard behind every wait: with a spin limit of a few polls a contended epoch must come back with an error, not hang
    (and the device must serve the next call)."""
    from yue_amd._shim import YueHipError
    data, ev_u, P0, Q0 = _problem(4000, 8, 6, 64, 5)
    dev.set_factors(P0, Q0)
    dev.set_interactions(data['indptr'], data['indices'], data['ev_ptr'], data['ev_i'])
    dev.set_option('epoch_exact', 1)
    dev.set_option('chain_spin', 2)
    try:
        with pytest.raises(YueHipError):
            dev.bpr_epoch(5, 0, 0, 0.02, 0.01, 0.01)
    finally:
        dev.set_option('chain_spin', 0)
        dev.set_option('epoch_exact', 0)
    dev.set_factors(P0, Q0)
    assert np.isfinite(dev.bpr_epoch(5, 0, 512, 0.02, 0.01, 0.01)[0])


def _exact_epoch(dev, data, P0, Q0, seed, split, fast, epochs=1, xcd=0):
    dev.set_factors(P0, Q0)
    dev.set_interactions(data['indptr'], data['indices'], data['ev_ptr'], data['ev_i'])
    dev.set_option('epoch_exact', 1)
    dev.set_option('chain_split', split)
    dev.set_option('chain_fast', fast)
    dev.set_option('chain_xcd', xcd)
    try:
        nll = [dev.bpr_epoch(seed, ep, 0, 0.02, 0.01, 0.01)[0] for ep in range(epochs)]
    finally:
        dev.set_option('epoch_exact', 0)
        dev.set_option('chain_split', -1)
        dev.set_option('chain_fast', 0)
        dev.set_option('chain_xcd', 0)
    return (nll,) + dev.get_factors()


@pytest.mark.parametrize('m,n,d,k', [(3000, 2000, 20, 128), (5000, 64, 12, 64), (700, 900, 70, 10), (400, 300, 25, 200)])
def test_wave_group_variant_is_bit_equal(dev, m, n, d, k):
    """Option chain_split: a run walked by a group of five waves (2 x loads / dependency chain / 2 x stores, hand-over through LDS) must produce
    exactly the factors (and the loss) of the one-wave kernel (70 events per user: runs longer than one segment of 64;
    5000 users on 64 items: the ring of packets is full of rows that wait for the same group's own stores) -- on all XCDs with
    write-through hand-offs, and with every working wave on ONE XCD and the rows handed over through its L2 (option chain_xcd)."""
    data, ev_u, P0, Q0 = _problem(m, n, d, k, 7)
    res = [_exact_epoch(dev, data, P0, Q0, 9, split, 0, epochs=2, xcd=xcd) for split, xcd in ((0, 0), (1, 0), (1, 1))]
    # (the loss is a sum of per-wave partials added in whatever order the waves finish)
    for other in res[1:]:
        assert all(abs(a - b) <= 1e-12 * abs(a) for a, b in zip(res[0][0], other[0]))
        assert np.array_equal(res[0][1], other[1]) and np.array_equal(res[0][2], other[2])
    print('one XCD: %d waves stayed' % dev.get_option('chain_last_waves'))


def test_wave_group_replay_of_an_interleaved_stream_is_bit_equal(dev):
    """The general stream (users come back: user rows versioned per run, skipped triplets, items repeated inside a run) through
    the wave-group kernel lands on the one-wave kernel's factors."""
    rs = np.random.RandomState(79)
    m, n, k, T = 300, 500, 128, 40000
    P0, Q0 = synth.init_factors(m, n, k, 80)
    u = rs.randint(0, m, size=T).astype(np.int32)
    u[2000:2300] = 11
    idx = np.arange(0, T - 1, 3)
    u[idx] = u[idx + 1]
    i = (n * rs.rand(T) ** 2).astype(np.int32)
    j = rs.randint(0, n, size=T).astype(np.int32)
    j[j == i] = (i[j == i] + 1) % n
    j[rs.rand(T) < 0.01] = -1
    res = []
    for split, xcd in ((0, 0), (1, 0), (1, 1)):
        dev.set_factors(P0, Q0)
        dev.set_option('chain_split', split)
        dev.set_option('chain_xcd', xcd)
        try:
            nll = dev.bpr_replay(u, i, j, 0.03, 0.02, 0.01)
        finally:
            dev.set_option('chain_split', -1)
            dev.set_option('chain_xcd', 0)
        res.append((nll,) + dev.get_factors())
    for other in res[1:]:
        assert abs(res[0][0] - other[0]) <= 1e-12 * abs(res[0][0]) and np.array_equal(res[0][1], other[1]) and np.array_equal(res[0][2], other[2])


@pytest.mark.parametrize('split', [0, 1])
@pytest.mark.parametrize('m,n,d,k', [(3000, 2000, 20, 128), (5000, 64, 12, 64), (700, 900, 70, 10), (400, 300, 25, 200), (20000, 5000, 20, 128)])
def test_fast_coefficient_stays_within_the_north_star_tolerance(dev, orc, split, m, n, d, k):
    """Option chain_fast: the step's coefficient c = fp32(lr (1 - sigmoid(x))) in single precision and the margin as ONE 64-lane
    sum of p . (qi - qj) -- not bit-equal to the reference's arithmetic, but within BASELINE.json's 1e-5 of the sequential
    oracle after two epochs (the loss is still summed from double-precision sigmoids of the margins)."""
    data, ev_u, P0, Q0 = _problem(m, n, d, k, 31 + k)
    dev.set_factors(P0, Q0)
    dev.set_interactions(data['indptr'], data['indices'], data['ev_ptr'], data['ev_i'])
    js = [dev.sample_negatives(5, ep) for ep in range(2)]
    nll, P, Q = _exact_epoch(dev, data, P0, Q0, 5, split, 1, epochs=2)
    Po, Qo = P0.copy(), Q0.copy()
    for ep in range(2):
        nll_o = orc.bpr_sequential(Po, Qo, ev_u, data['ev_i'], js[ep], 0.02, 0.01, 0.01)
        assert abs(nll[ep] - nll_o) <= 1e-6 * abs(nll_o)
    print('fast coefficient (split %d): rel P %.2e Q %.2e; bit-equal to the oracle: P %.4f Q %.4f' % (split, rel_err(P, Po), rel_err(Q, Qo), np.mean(P == Po), np.mean(Q == Qo)))
    assert rel_err(P, Po) < 1e-5 and rel_err(Q, Qo) < 1e-5
