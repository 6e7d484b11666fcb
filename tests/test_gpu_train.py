"""GPU parity of the training path, through the C ABI (yue_amd/_shim.py -> libyue_hip.so).

Checker = oracle/ (pinned to the reference by tests/test_oracle_golden.py) and the reference's own
golden outputs.  Tolerances: BASELINE.json north_star -- factor matrices within 1e-5 rel fp32
(rel = max|a-b| / max|b|); integer work (sampler) bit-exact.
"""
import numpy as np
import pytest

from yue_amd import synth
from yue_amd.dist import epoch_round_ptr
from util import gz, rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-5


_SEQ = [1]


@pytest.fixture(scope='module', params=[1, 0], ids=['user_rows_sequential', 'user_rows_per_round'], autouse=True)
def user_rows(request):
    """The whole module runs twice: with the epoch path's default on one GPU -- a wave owns a user and walks the user's events in
    order (k_round_u; oracle: orc_bpr_rounds_seq_user) -- and with user rows under round semantics (option round_user_seq = 0:
    k_round_m + dP, the form a communicator runs; oracle: orc_bpr_rounds).  Every Device of the module gets the option."""
    from yue_amd import _shim
    orig = _shim.Device.__init__

    def init(self, *a, **k):
        orig(self, *a, **k)
        self.set_option('round_user_seq', request.param)
        self.set_option('round_fast', 0)      # the reference's double-precision sigmoid in k_round_u too: the tight bounds of this module hold (the default, a single-precision coefficient, has a test of its own below)
    _shim.Device.__init__ = init
    _SEQ[0] = request.param
    yield request.param
    _shim.Device.__init__ = orig
    _SEQ[0] = 1


def _epoch_oracle(orc, meta=1):
    """the restatement of what yue_bpr_epoch runs on one GPU under the module's current option (round_meta = 0, the kernel that
    counts and retires inside the round launches, keeps user rows under round semantics)"""
    return orc.bpr_rounds_seq_user if _SEQ[0] and meta else orc.bpr_rounds


def _heavy():
    if not _SEQ[0]:
        pytest.skip('BASELINE-size case: run once, under the default')


@pytest.fixture(scope='module')
def dev(user_rows):
    from yue_amd._shim import Device
    d = Device(0, raise_errors=True)
    yield d
    d.close()


def _golden(tag):
    z = gz('g4_%s.npz' % tag)
    iters = int(z['iters'])
    return z, iters, len(z['u']) // iters


@pytest.mark.parametrize('tag', ['c1_k10_e1', 'd2_k64_e1', 'd3_k128_e2'])
def test_replay_matches_reference_and_oracle(dev, orc, tag):
    # BPR.py:42-58 on the reference's own triplet stream (lRate stays 0.02 over these epochs)
    z, iters, E = _golden(tag)
    m, n, k, seed = int(z['m']), int(z['n']), int(z['k']), int(z['seed'])
    P0, Q0 = synth.init_factors(m, n, k, seed)
    Po, Qo = P0.copy(), Q0.copy()
    dev.set_factors(P0, Q0)
    lr = 0.02
    nll_g = nll_o = 0.0
    for ep in range(iters):
        sl = slice(ep * E, (ep + 1) * E)
        nll_g = dev.bpr_replay(z['u'][sl], z['i'][sl], z['j'][sl], lr, 0.01, 0.01)
        nll_o = orc.bpr_sequential(Po, Qo, z['u'][sl], z['i'][sl], z['j'][sl], lr, 0.01, 0.01)
        assert abs(nll_g - nll_o) <= 1e-9 * abs(nll_o)
    P, Q = dev.get_factors()
    # same dot order as the oracle: differences can only come from exp/log last-bit rounding
    assert rel_err(P, Po) < 1e-6 and rel_err(Q, Qo) < 1e-6
    print('bit-exact vs oracle: P %.4f Q %.4f' % (np.mean(P == Po), np.mean(Q == Qo)))
    # and the reference's own result
    assert rel_err(P, z['P']) < TOL and rel_err(Q, z['Q']) < TOL
    sp, sq = dev.sumsq()
    assert abs(sp - orc.sumsq(P)) <= 1e-12 * sp and abs(sq - orc.sumsq(Q)) <= 1e-12 * sq
    loss = np.float32(nll_g) + (np.float32(0.01) * np.float32(sp) + np.float32(0.01) * np.float32(sq))
    assert abs(float(loss) - float(z['loss'])) <= TOL * abs(float(z['loss']))


def test_replay_edge_cases(dev, orc):
    rs = np.random.RandomState(1)
    m, n, k = 7, 9, 33                       # k not a multiple of 32, tiny shapes
    P0 = rs.rand(m, k).astype(np.float32) / 10
    Q0 = rs.rand(n, k).astype(np.float32) / 10
    u = np.array([0, 0, 0, 3, 6, 0, 3], np.int32)
    i = np.array([1, 1, 2, 8, 0, 1, 8], np.int32)
    j = np.array([2, 4, 1, 0, -1, 5, 7], np.int32)   # one skipped triplet, repeated rows, i/j swapped roles
    dev.set_factors(P0, Q0)
    nll = dev.bpr_replay(u, i, j, 0.05, 0.02, 0.03)
    Po, Qo = P0.copy(), Q0.copy()
    nll_o = orc.bpr_sequential(Po, Qo, u, i, j, 0.05, 0.02, 0.03)
    P, Q = dev.get_factors()
    assert rel_err(P, Po) < 1e-6 and rel_err(Q, Qo) < 1e-6 and abs(nll - nll_o) < 1e-9 * nll_o
    # empty stream is a no-op
    assert dev.bpr_replay(u[:0], i[:0], j[:0], 0.05, 0.02, 0.03) == 0.0
    P2, Q2 = dev.get_factors()
    assert np.array_equal(P, P2) and np.array_equal(Q, Q2)
    from yue_amd._shim import YueHipError
    with pytest.raises(YueHipError):
        dev.bpr_replay(np.array([7], np.int32), np.array([0], np.int32), np.array([1], np.int32), 0.05, 0.02, 0.03)


@pytest.mark.parametrize('tag,W', [('c1_k10_e1', 64), ('c1_k10_e1', 1000), ('d2_k64_e1', 257), ('d3_k128_e2', 4096)])
def test_rounds_match_oracle(dev, orc, tag, W):
    z, iters, E = _golden(tag)
    m, n, k = int(z['m']), int(z['n']), int(z['k'])
    P0, Q0 = synth.init_factors(m, n, k, 11)
    T = E
    rp = np.unique(np.concatenate([np.arange(0, T, W), [T]])).astype(np.int64)
    dev.set_factors(P0, Q0)
    nll = dev.bpr_rounds(z['u'][:T], z['i'][:T], z['j'][:T], rp, 0.02, 0.01, 0.01)
    Po, Qo = P0.copy(), Q0.copy()
    nll_o = orc.bpr_rounds(Po, Qo, z['u'][:T], z['i'][:T], z['j'][:T], rp, 0.02, 0.01, 0.01)
    P, Q = dev.get_factors()
    assert rel_err(P, Po) < TOL and rel_err(Q, Qo) < TOL
    assert abs(nll - nll_o) <= 1e-9 * abs(nll_o)


def test_rounds_with_arbitrary_user_order_and_wide_factors(dev, orc):
    # explicit triplets whose users are NOT grouped (runs of equal users are short and repeat), k = 200
    # (four registers per lane and row, the 4-events-per-wave instance), skipped triplets, ragged rounds
    rs = np.random.RandomState(21)
    m, n, k, T = 700, 900, 200, 20000
    P0 = rs.rand(m, k).astype(np.float32) / 10
    Q0 = rs.rand(n, k).astype(np.float32) / 10
    u = rs.randint(0, m, size=T).astype(np.int32)
    u[::3] = u[1::3][:len(u[::3])]                      # some consecutive repeats
    i = rs.randint(0, n, size=T).astype(np.int32)
    j = ((i + 1 + rs.randint(0, n - 1, size=T)) % n).astype(np.int32)
    j[rs.rand(T) < 0.01] = -1
    rp = np.array([0, 1, 2, 50, 51, 4000, 4003, 12345, T], np.int64)
    dev.set_factors(P0, Q0)
    nll = dev.bpr_rounds(u, i, j, rp, 0.03, 0.02, 0.01)
    Po, Qo = P0.copy(), Q0.copy()
    nll_o = orc.bpr_rounds(Po, Qo, u, i, j, rp, 0.03, 0.02, 0.01)
    P, Q = dev.get_factors()
    assert rel_err(P, Po) < TOL and rel_err(Q, Qo) < TOL and abs(nll - nll_o) <= 1e-9 * abs(nll_o)
    # the same stream replayed exactly (sequential semantics) on the wide factors
    dev.set_factors(P0, Q0)
    nll = dev.bpr_replay(u[:3000], i[:3000], j[:3000], 0.03, 0.02, 0.01)
    Po, Qo = P0.copy(), Q0.copy()
    nll_o = orc.bpr_sequential(Po, Qo, u[:3000], i[:3000], j[:3000], 0.03, 0.02, 0.01)
    P, Q = dev.get_factors()
    assert rel_err(P, Po) < 1e-6 and rel_err(Q, Qo) < 1e-6 and abs(nll - nll_o) <= 1e-9 * abs(nll_o)


def test_rounds_of_one_reduce_to_the_reference_loop(dev):
    # S-round with one triplet per round is the sequential loop: compare with the reference's output
    z, iters, E = _golden('d2_k64_e1')
    m, n, k, seed = int(z['m']), int(z['n']), int(z['k']), int(z['seed'])
    P0, Q0 = synth.init_factors(m, n, k, seed)
    dev.set_factors(P0, Q0)
    T = E
    dev.bpr_rounds(z['u'][:T], z['i'][:T], z['j'][:T], np.arange(T + 1), 0.02, 0.01, 0.01)
    P, Q = dev.get_factors()
    assert rel_err(P, z['P']) < TOL and rel_err(Q, z['Q']) < TOL


def _synth_problem(m, n, d, k, seed=5):
    data = synth.make_arrays(m, n, d, seed=seed)
    P0, Q0 = synth.init_factors(m, n, k, seed)
    ev_u = np.repeat(np.arange(m, dtype=np.int32), np.diff(data['ev_ptr']))
    return data, P0, Q0, ev_u


@pytest.mark.parametrize('m,n,d,k,W', [(300, 400, 20, 10, 128), (500, 257, 30, 64, 1000), (2000, 3000, 50, 128, 8192)])
def test_fused_epoch_matches_oracle(dev, orc, m, n, d, k, W):
    data, P0, Q0, ev_u = _synth_problem(m, n, d, k)
    dev.set_factors(P0, Q0)
    dev.set_interactions(data['indptr'], data['indices'], data['ev_ptr'], data['ev_i'])
    seed = 987654321
    Po, Qo = P0.copy(), Q0.copy()
    E = len(ev_u)
    rp = np.array(epoch_round_ptr(data['ev_ptr'], W), np.int64)
    for epoch in range(2):
        # integer work: the device sampler's negatives are bit-exact with the oracle's
        j_gpu = dev.sample_negatives(seed, epoch)
        j_orc = orc.sample_counter(seed, epoch, ev_u, n, data['indptr'], data['indices'])
        assert np.array_equal(j_gpu, j_orc)
        nll, sp, sq = dev.bpr_epoch(seed, epoch, W, 0.02, 0.01, 0.01)
        nll_o = _epoch_oracle(orc)(Po, Qo, ev_u, data['ev_i'], j_orc, rp, 0.02, 0.01, 0.01)
        P, Q = dev.get_factors()
        assert rel_err(P, Po) < TOL and rel_err(Q, Qo) < TOL
        assert abs(nll - nll_o) <= 1e-9 * abs(nll_o)
        assert abs(sp - orc.sumsq(P)) <= 1e-12 * sp and abs(sq - orc.sumsq(Q)) <= 1e-12 * sq


@pytest.mark.parametrize('tpw', [2, 4, 8, 16])
def test_round_kernel_under_heavy_contention_and_many_wave_passes(orc, tpw):
    # few items, big rounds: every item row is contended (about 40 touches per row and round, hot rows
    # several hundred), and the grid has more waves than the chip holds at once (late waves must still
    # see the round's touch totals, not the retired counts).  The fp32 order of the atomic sums is the
    # only freedom left; measured 1e-6 .. 6e-6 here, against the 1e-5 bar.
    from yue_amd._shim import Device
    m, n, d, k, W = 30000, 3000, 20, 64, 65536
    data, P0, Q0, ev_u = _synth_problem(m, n, d, k, seed=77)
    dev = Device(0, raise_errors=True)
    dev.set_option('round_tpw', tpw)
    dev.set_factors(P0, Q0)
    dev.set_interactions(data['indptr'], data['indices'], data['ev_ptr'], data['ev_i'])
    Po, Qo = P0.copy(), Q0.copy()
    E = len(ev_u)
    rp = np.array(epoch_round_ptr(data['ev_ptr'], W), np.int64)
    for epoch in range(2):
        j = orc.sample_counter(5, epoch, ev_u, n, data['indptr'], data['indices'])
        nll, _, _ = dev.bpr_epoch(5, epoch, W, 0.01, 0.01, 0.01)
        nll_o = _epoch_oracle(orc)(Po, Qo, ev_u, data['ev_i'], j, rp, 0.01, 0.01, 0.01)
        P, Q = dev.get_factors()
        assert rel_err(P, Po) < TOL and rel_err(Q, Qo) < TOL
        assert abs(nll - nll_o) <= 1e-9 * abs(nll_o)
    dev.close()


def test_tuning_knobs_do_not_change_the_epoch(orc):
    # round_stage: staging rows vs float atomics for rows with 2..4 touches -> same sums, another fp32 order;
    # round_tpw: events per wave
    from yue_amd._shim import Device
    m, n, d, k, W = 4000, 2500, 25, 128, 8192
    data, P0, Q0, ev_u = _synth_problem(m, n, d, k, seed=21)
    E = len(ev_u)
    rp = np.array(epoch_round_ptr(data['ev_ptr'], W), np.int64)
    j = orc.sample_counter(9, 0, ev_u, n, data['indptr'], data['indices'])
    ref = {}
    for meta_ in (1, 0):
        Po, Qo = P0.copy(), Q0.copy()
        ref[meta_] = (_epoch_oracle(orc, meta_)(Po, Qo, ev_u, data['ev_i'], j, rp, 0.02, 0.01, 0.01), Po, Qo)
    # (with round_meta 1 also 2..64: staged up to that many touches per row -- W = 8,192 events on 2,500 items: 6.5
    # touches per row and round on average, blocks of every size up to 64 occur);
    # round_meta: touch metadata from the per-epoch pre-pass + fold launches (default) vs touches counted and contended
    # rows finished inside the round launches
    for tpw, stage, meta in [(0, 1, 1), (0, 0, 1), (4, 1, 1), (2, 0, 1), (0, 2, 1), (8, 3, 1), (0, 7, 1), (4, 13, 1), (0, 64, 1), (0, 1, 0), (0, 0, 0), (4, 1, 0)]:
        dev = Device(0, raise_errors=True)
        dev.set_option('round_tpw', tpw)
        dev.set_option('round_stage', stage)
        dev.set_option('round_meta', meta)
        dev.set_factors(P0, Q0)
        dev.set_interactions(data['indptr'], data['indices'], data['ev_ptr'], data['ev_i'])
        nll, _, _ = dev.bpr_epoch(9, 0, W, 0.02, 0.01, 0.01)
        P, Q = dev.get_factors()
        dev.close()
        nll_o, Po, Qo = ref[meta]
        assert rel_err(P, Po) < TOL and rel_err(Q, Qo) < TOL and abs(nll - nll_o) <= 1e-9 * abs(nll_o), (tpw, stage, meta)


def test_staged_block_size_follows_the_rounds_mean_touches_per_row():
    # default round_stage: largest staged block = ceil(2 x mean touches per item row) at k > 64, 4 x at k <= 64, within 4..64
    # (mean = 2 x events of the widest round / item rows; rounds are blocks of whole users: 328 users x 25 events here)
    from yue_amd._shim import Device
    m, n, d, W = 4000, 2500, 25, 8192
    for k, want in ((128, 14), (64, 27), (200, 14)):
        data, P0, Q0, ev_u = _synth_problem(m, n, d, k, seed=21)
        dev = Device(0, raise_errors=True)
        dev.set_factors(P0, Q0)
        dev.set_interactions(data['indptr'], data['indices'], data['ev_ptr'], data['ev_i'])
        dev.bpr_epoch(9, 0, W, 0.02, 0.01, 0.01)
        assert dev.get_option('round_last_stage_max') == want, (k, dev.get_option('round_last_stage_max'))
        dev.bpr_epoch(9, 1, 64, 0.02, 0.01, 0.01)             # rounds of 3 users = 75 events on 2,500 rows: the floor of 4
        assert dev.get_option('round_last_stage_max') == 4
        dev.set_option('round_stage', 9)
        dev.bpr_epoch(9, 2, W, 0.02, 0.01, 0.01)
        assert dev.get_option('round_last_stage_max') == 9
        dev.close()


def test_epoch_metadata_pre_pass_over_several_item_ranges(orc):
    # n above one LDS range of the pre-pass (37,888 item rows): three ranges per round, staging blocks of one round handed
    # out by several work items; a popular head (rows with hundreds of touches per round -> float atomics), rows with 2..4
    # touches (staged) and single touches side by side.  Both paths against the oracle.
    from yue_amd._shim import Device
    m, n, d, k, W = 12000, 100000, 40, 32, 16384
    data, P0, Q0, ev_u = _synth_problem(m, n, d, k, seed=5)
    rp = np.array(epoch_round_ptr(data['ev_ptr'], W), np.int64)
    refs = {}
    for meta in (1, 0):
        Po, Qo = P0.copy(), Q0.copy()
        ref = []
        for epoch in range(2):
            j = orc.sample_counter(13, epoch, ev_u, n, data['indptr'], data['indices'])
            ref.append(_epoch_oracle(orc, meta)(Po, Qo, ev_u, data['ev_i'], j, rp, 0.03, 0.01, 0.01))
        refs[meta] = (ref, Po, Qo)
    touched = np.bincount(np.concatenate([data['ev_i'][:W], j[:W]]), minlength=n)
    assert (touched == 1).sum() > 1000 and ((touched >= 2) & (touched <= 4)).sum() > 1000 and (touched > 4).sum() > 30
    for meta in (1, 0):
        dev = Device(0, raise_errors=True)
        dev.set_option('round_meta', meta)
        ref, Po, Qo = refs[meta]
        dev.set_factors(P0, Q0)
        dev.set_interactions(data['indptr'], data['indices'], data['ev_ptr'], data['ev_i'])
        for epoch in range(2):
            nll, _, _ = dev.bpr_epoch(13, epoch, W, 0.03, 0.01, 0.01)
            assert abs(nll - ref[epoch]) <= 1e-9 * abs(ref[epoch]), (meta, epoch)
        P, Q = dev.get_factors()
        dev.close()
        assert rel_err(P, Po) < TOL and rel_err(Q, Qo) < TOL, meta


def test_epoch_path_on_a_large_catalogue(orc):
    _heavy()
    # more item rows than the plain pre-pass takes (454,656): a round's touches are bucketed by item range first
    # (k_round_bucket), 19 ranges of 32,768 rows here; sparse touches (most rows untouched) and a popular head
    from yue_amd._shim import Device
    m, n, d, k, W = 20000, 600000, 25, 16, 30000
    data, P0, Q0, ev_u = _synth_problem(m, n, d, k, seed=8)
    rp = np.array(epoch_round_ptr(data['ev_ptr'], W), np.int64)
    dev = Device(0, raise_errors=True)
    dev.set_factors(P0, Q0)
    dev.set_interactions(data['indptr'], data['indices'], data['ev_ptr'], data['ev_i'])
    assert dev.get_option('round_path') == 1
    Po, Qo = P0.copy(), Q0.copy()
    for epoch in range(2):
        j = orc.sample_counter(21, epoch, ev_u, n, data['indptr'], data['indices'])
        nll, _, _ = dev.bpr_epoch(21, epoch, W, 0.03, 0.01, 0.01)
        nll_o = _epoch_oracle(orc)(Po, Qo, ev_u, data['ev_i'], j, rp, 0.03, 0.01, 0.01)
        P, Q = dev.get_factors()
        assert rel_err(P, Po) < TOL and rel_err(Q, Qo) < TOL and abs(nll - nll_o) <= 1e-9 * abs(nll_o), epoch
    dev.close()


def test_default_round_size_is_one_resident_wave_set(orc):
    # round_events = 0 -> yue_default_round_events: a multiple of 1024 that fits the chip's resident workgroups
    # (32 events each at 8 events per wave); the epoch equals the explicit call
    from yue_amd._shim import Device
    m, n, d, k = 3000, 2000, 20, 128
    data, P0, Q0, ev_u = _synth_problem(m, n, d, k, seed=11)
    out = []
    for explicit in (False, True):
        dev = Device(0, raise_errors=True)
        dev.set_factors(P0, Q0)
        dev.set_interactions(data['indptr'], data['indices'], data['ev_ptr'], data['ev_i'])
        W = dev.default_round_events()
        assert W % 1024 == 0 and 8192 <= W <= 131072
        nll, _, _ = dev.bpr_epoch(3, 0, W if explicit else 0, 0.02, 0.01, 0.01)
        out.append((nll,) + dev.get_factors())
        dev.close()
    assert abs(out[0][0] - out[1][0]) <= 1e-9 * abs(out[1][0]) and rel_err(out[0][1], out[1][1]) < TOL and rel_err(out[0][2], out[1][2]) < TOL
    E = len(ev_u)
    j = orc.sample_counter(3, 0, ev_u, n, data['indptr'], data['indices'])
    rp = np.array(epoch_round_ptr(data['ev_ptr'], W), np.int64)
    Po, Qo = P0.copy(), Q0.copy()
    nll_o = _epoch_oracle(orc)(Po, Qo, ev_u, data['ev_i'], j, rp, 0.02, 0.01, 0.01)
    assert rel_err(out[0][1], Po) < TOL and rel_err(out[0][2], Qo) < TOL and abs(out[0][0] - nll_o) <= 1e-9 * abs(nll_o)


def test_default_round_size_grows_with_the_item_count():
    # epoch path: up to 6 resident sets of the update kernel, as long as a round holds at most four events per item row
    from yue_amd._shim import Device
    k = 128
    sizes = {}
    for n in (1000, 20000, 130000, 200000, 400000):
        dev = Device(0, raise_errors=True)
        dev.set_factors(np.zeros((8, k), np.float32), np.zeros((n, k), np.float32))
        sizes[n] = dev.default_round_events()
        assert dev.get_option('round_path') == 1
        dev.set_option('round_meta', 0)                      # the kernel that finishes contended rows inside the launch: one set of its own
        one_old = dev.default_round_events()
        assert dev.get_option('round_path') == 0 and one_old % 1024 == 0 and one_old <= sizes[1000]
        dev.close()
    one = sizes[1000]
    assert one % 1024 == 0
    for n in (20000, 130000, 200000, 400000):
        sets = min(6, max(1, 4 * n // one))
        assert sizes[n] == one * (min(sets, 4) if one * sets < n else sets), (n, sizes)     # (wide catalogues: at most 4 sets)
    assert sizes[400000] == 4 * one and sizes[200000] == 6 * one
    dev = Device(0, raise_errors=True)                           # large catalogues: bucketed pre-pass, still the epoch path ...
    dev.set_factors(np.zeros((8, 4), np.float32), np.zeros((500000, 4), np.float32))
    assert dev.get_option('round_path') == 1
    dev.set_factors(np.zeros((8, 1), np.float32), np.zeros((9000000, 1), np.float32))     # ... up to 256 ranges of 32,768 rows
    assert dev.get_option('round_path') == 0
    dev.close()


def test_fused_epoch_skips_unsampleable_and_empty_users(dev, orc):
    # user 0 listened to all but one item (sampler mostly rejects), user 1 has no events at all
    m, n, k = 3, 40, 16
    rows = [np.arange(39), np.zeros(0, int), np.array([3, 7])]
    indptr = np.cumsum([0] + [len(r) for r in rows]).astype(np.int64)
    indices = np.concatenate(rows).astype(np.int32)
    ev_ptr = np.array([0, 39, 39, 41], np.int64)
    ev_i = np.concatenate([np.arange(39), [3, 7]]).astype(np.int32)
    ev_u = np.repeat(np.arange(3, dtype=np.int32), np.diff(ev_ptr))
    P0, Q0 = synth.init_factors(m, n, k, 3)
    dev.set_factors(P0, Q0)
    dev.set_interactions(indptr, indices, ev_ptr, ev_i)
    j_gpu = dev.sample_negatives(77, 0)
    j_orc = orc.sample_counter(77, 0, ev_u, n, indptr, indices)
    assert np.array_equal(j_gpu, j_orc)
    assert set(j_orc[:39].tolist()) <= {39, -1}
    nll, _, _ = dev.bpr_epoch(77, 0, 16, 0.02, 0.01, 0.01)
    Po, Qo = P0.copy(), Q0.copy()
    rp = np.array(epoch_round_ptr(ev_ptr, 16), np.int64)
    nll_o = _epoch_oracle(orc)(Po, Qo, ev_u, ev_i, j_orc, rp, 0.02, 0.01, 0.01)
    P, Q = dev.get_factors()
    assert rel_err(P, Po) < TOL and rel_err(Q, Qo) < TOL and abs(nll - nll_o) <= 1e-9 * max(abs(nll_o), 1e-30)
    assert np.array_equal(P[1], P0[1])


def test_user_factors_beyond_two_gib(orc):
    _heavy()
    # BASELINE config 4 keeps 10M x 128 user factors (5 GB) on every GPU: the round kernel addresses P and
    # dP with 31-bit offsets relative to the smallest user of a wave's batch.  Here 4.3M users x 128 = 2.2 GB.
    from yue_amd._shim import Device
    m, n, d, k, W = 4300000, 2000, 2, 128, 32768
    data = synth.make_arrays(m, n, d, seed=31)
    rng = np.random.default_rng(32)
    P0 = rng.random((m, k), dtype=np.float32) / 10
    Q0 = rng.random((n, k), dtype=np.float32) / 10
    assert P0.nbytes > (1 << 31)
    ev_u = np.repeat(np.arange(m, dtype=np.int32), np.diff(data['ev_ptr']))
    E = len(ev_u)
    dev = Device(0, raise_errors=True)
    try:
        dev.set_factors(P0, Q0)
        dev.set_interactions(data['indptr'], data['indices'], data['ev_ptr'], data['ev_i'])
        j = orc.sample_counter(6, 0, ev_u, n, data['indptr'], data['indices'])
        nll, sp, _ = dev.bpr_epoch(6, 0, W, 0.02, 0.01, 0.01)
        P, Q = dev.get_factors()
        Po, Qo = P0.copy(), Q0.copy()
        rp = np.array(epoch_round_ptr(data['ev_ptr'], W), np.int64)
        nll_o = _epoch_oracle(orc)(Po, Qo, ev_u, data['ev_i'], j, rp, 0.02, 0.01, 0.01)
        assert rel_err(P, Po) < TOL and rel_err(Q, Qo) < TOL and abs(nll - nll_o) <= 1e-9 * abs(nll_o)
        assert rel_err(P[-1000:], Po[-1000:]) < TOL and not np.array_equal(P[-1000:], P0[-1000:])
        assert abs(sp - orc.sumsq(P)) <= 1e-12 * sp
        # explicit rounds over the last users (offsets above 2 GiB), ungrouped order inside the rounds
        T = 40000
        perm = np.random.RandomState(1).permutation(T)
        u2, i2, j2 = ev_u[E - T:][perm], data['ev_i'][E - T:][perm], j[E - T:][perm]
        rp2 = np.array([0, 15000, T], np.int64)
        dev.set_factors(P0, Q0)
        nll2 = dev.bpr_rounds(u2, i2, j2, rp2, 0.02, 0.01, 0.01)
        P, Q = dev.get_factors()
        Po[:] = P0
        Qo[:] = Q0
        nll2_o = orc.bpr_rounds(Po, Qo, u2, i2, j2, rp2, 0.02, 0.01, 0.01)
        assert rel_err(P, Po) < TOL and rel_err(Q, Qo) < TOL and abs(nll2 - nll2_o) <= 1e-9 * abs(nll2_o)
    finally:
        dev.close()


def test_full_size_properties(dev):
    _heavy()
    # BASELINE config 2 shape (100K x 50K, k=64): size-independent properties instead of the oracle
    m, n, d, k = 100000, 50000, 50, 64
    data, P0, Q0, ev_u = _synth_problem(m, n, d, k, seed=20260001)
    dev.set_factors(P0, Q0)
    dev.set_interactions(data['indptr'], data['indices'], data['ev_ptr'], data['ev_i'])
    E = len(ev_u)
    j = dev.sample_negatives(1, 0)
    assert (j >= 0).all() and (j < n).all()
    # no sampled negative is a listened item
    key = np.repeat(np.arange(m, dtype=np.int64), np.diff(data['indptr'])) * n + data['indices']
    assert not np.isin(ev_u.astype(np.int64) * n + j, key).any()
    losses = []
    for epoch in range(3):
        nll, sp, sq = dev.bpr_epoch(1, epoch, 32768, 0.05, 0.01, 0.01)
        assert np.isfinite(nll) and np.isfinite(sp) and np.isfinite(sq)
        losses.append(nll)
    assert losses[2] < losses[1] < losses[0] < E * np.log(2) * 1.01
    # a second context replays to the same result (atomics only reorder fp32 sums)
    P, Q = dev.get_factors()
    from yue_amd._shim import Device
    d2 = Device(0, raise_errors=True)
    d2.set_factors(P0, Q0)
    d2.set_interactions(data['indptr'], data['indices'], data['ev_ptr'], data['ev_i'])
    for epoch in range(3):
        d2.bpr_epoch(1, epoch, 32768, 0.05, 0.01, 0.01)
    P2, Q2 = d2.get_factors()
    d2.close()
    assert rel_err(P2, P) < TOL and rel_err(Q2, Q) < TOL


@pytest.mark.parametrize('case,k,W', [(0, 200, 1), (1, 256, 37), (2, 192, 5), (3, 100, 100000), (4, 33, 100000), (5, 100, 1), (6, 7, 1),
                                      (7, 33, 37), (8, 128, 4096), (9, 7, 256), (10, 3, 4096), (11, 64, 1000), (12, 16, 5),
                                      (13, 16, 3000), (14, 64, 20000), (15, 200, 700)])
def test_epoch_path_on_random_shapes(orc, case, k, W):
    # random small problems through the epoch path (pre-pass metadata, update launches, fold launches): odd and wide k (all
    # three register layouts), rounds from single users (a handful of events: fold lists shorter than a wave's group) to more
    # than the epoch, users without events, a skewed item popularity (staged and hot rows in almost every round).
    # (case 0 is the shape that exposed a fold-list bug of the wide-k kernel in round 2: tools/experiments/random_shapes_probe.py)
    from yue_amd._shim import Device
    rs = np.random.RandomState(1000 + case)
    m = int(rs.randint(1, 1500)); n = int(rs.randint(50, 4000))
    if case >= 13:
        n = int(rs.randint(38000, 120000))                   # several item ranges in the pre-pass (one LDS word per row, 37,888 per range)
    P0 = rs.rand(m, k).astype(np.float32) / 10
    Q0 = rs.rand(n, k).astype(np.float32) / 10
    pop = 1.0 / (np.arange(n) + 3.0) ** 0.9
    pop = pop[rs.permutation(n)]; pop /= pop.sum()
    ev, rows = [], []
    for u in range(m):
        cnt = int(rs.randint(0, min(60, n - 1) + 1)) if rs.rand() > 0.1 else 0       # some users have no events
        it = rs.choice(n, size=cnt, p=pop) if cnt else np.zeros(0, np.int64)
        ev.append(it.astype(np.int32)); rows.append(np.unique(it).astype(np.int32))
    ev_ptr = np.cumsum([0] + [len(e) for e in ev]).astype(np.int64)
    if ev_ptr[-1] == 0:
        pytest.skip('no events drawn')
    ev_i = np.concatenate(ev).astype(np.int32)
    indptr = np.cumsum([0] + [len(r) for r in rows]).astype(np.int64)
    indices = np.concatenate(rows).astype(np.int32)
    ev_u = np.repeat(np.arange(m, dtype=np.int32), np.diff(ev_ptr))
    rp = np.array(epoch_round_ptr(ev_ptr, W), np.int64)
    for meta, bucket in ((1, 0), (1, 1), (0, 0)):            # the epoch path (plain / bucketed pre-pass), and the kernel that counts and retires inside the launch
        dev = Device(0, raise_errors=True)
        dev.set_option('round_meta', meta)
        dev.set_option('round_bucket', bucket)
        dev.set_factors(P0, Q0)
        dev.set_interactions(indptr, indices, ev_ptr, ev_i)
        assert dev.get_option('round_path') == meta
        Po, Qo = P0.copy(), Q0.copy()
        for epoch in range(2):
            j = orc.sample_counter(77, epoch, ev_u, n, indptr, indices)
            nll, sp, sq = dev.bpr_epoch(77, epoch, W, 0.03, 0.01, 0.02)
            nll_o = _epoch_oracle(orc, meta)(Po, Qo, ev_u, ev_i, j, rp, 0.03, 0.01, 0.02)
            P, Q = dev.get_factors()
            assert rel_err(P, Po) < TOL and rel_err(Q, Qo) < TOL, (case, m, n, k, W, epoch, meta, bucket)
            assert abs(nll - nll_o) <= 1e-8 * max(abs(nll_o), 1e-30), (case, m, n, k, W, epoch, meta, bucket)
        dev.close()


def test_one_device_serves_problems_of_different_sizes_in_turn(dev, orc):
    # the epoch path's buffers (metadata, fold lists, round boundaries, staging rows) follow the uploaded problem
    for (m, n, d, k, W, seed) in [(300, 500, 10, 32, 256, 1), (2500, 3000, 40, 128, 20000, 2), (50, 4000, 5, 200, 64, 3), (2500, 700, 30, 64, 4096, 4)]:
        data, P0, Q0, ev_u = _synth_problem(m, n, d, k, seed=seed)
        dev.set_factors(P0, Q0)
        dev.set_interactions(data['indptr'], data['indices'], data['ev_ptr'], data['ev_i'])
        Po, Qo = P0.copy(), Q0.copy()
        rp = np.array(epoch_round_ptr(data['ev_ptr'], W), np.int64)
        for epoch in range(2):
            j = orc.sample_counter(6, epoch, ev_u, n, data['indptr'], data['indices'])
            nll, _, _ = dev.bpr_epoch(6, epoch, W, 0.02, 0.01, 0.01)
            nll_o = _epoch_oracle(orc)(Po, Qo, ev_u, data['ev_i'], j, rp, 0.02, 0.01, 0.01)
            P, Q = dev.get_factors()
            assert rel_err(P, Po) < TOL and rel_err(Q, Qo) < TOL and abs(nll - nll_o) <= 1e-9 * abs(nll_o), (m, n, k, W, epoch)


def test_round_size_may_change_between_epochs_on_one_device(dev, orc):
    # the pre-pass keeps the round boundaries of the last call on the device: another round size re-uploads them
    m, n, d, k = 900, 1200, 25, 64
    data, P0, Q0, ev_u = _synth_problem(m, n, d, k, seed=31)
    dev.set_factors(P0, Q0)
    dev.set_interactions(data['indptr'], data['indices'], data['ev_ptr'], data['ev_i'])
    Po, Qo = P0.copy(), Q0.copy()
    for epoch, W in enumerate([1000, 5000, 1000, 64, 5000]):
        rp = np.array(epoch_round_ptr(data['ev_ptr'], W), np.int64)
        j = orc.sample_counter(4, epoch, ev_u, n, data['indptr'], data['indices'])
        nll, _, _ = dev.bpr_epoch(4, epoch, W, 0.02, 0.01, 0.01)
        nll_o = _epoch_oracle(orc)(Po, Qo, ev_u, data['ev_i'], j, rp, 0.02, 0.01, 0.01)
        P, Q = dev.get_factors()
        assert rel_err(P, Po) < TOL and rel_err(Q, Qo) < TOL and abs(nll - nll_o) <= 1e-9 * abs(nll_o), (epoch, W)


@pytest.mark.parametrize('m,n,k', [(1, 3, 1), (2, 2, 3), (5, 40, 16), (3, 70, 256)])
def test_degenerate_shapes(orc, m, n, k):
    # smallest and widest supported shapes through every entry point
    from yue_amd._shim import Device
    rs = np.random.RandomState(m * 100 + n + k)
    P0 = rs.rand(m, k).astype(np.float32) / 10
    Q0 = rs.rand(n, k).astype(np.float32) / 10
    per_user = [np.sort(rs.choice(n, size=rs.randint(1, n), replace=False)) for _ in range(m)]
    ev = [rs.permutation(np.repeat(r, 2))[:max(1, len(r))] for r in per_user]
    ev_ptr = np.cumsum([0] + [len(e) for e in ev]).astype(np.int64)
    ev_i = np.concatenate(ev).astype(np.int32)
    indptr = np.cumsum([0] + [len(r) for r in per_user]).astype(np.int64)
    indices = np.concatenate(per_user).astype(np.int32)
    ev_u = np.repeat(np.arange(m, dtype=np.int32), np.diff(ev_ptr))
    dev = Device(0, raise_errors=True)
    dev.set_factors(P0, Q0)
    dev.set_interactions(indptr, indices, ev_ptr, ev_i)
    j = dev.sample_negatives(9, 0)
    assert np.array_equal(j, orc.sample_counter(9, 0, ev_u, n, indptr, indices))
    nll, sp, sq = dev.bpr_epoch(9, 0, 3, 0.05, 0.01, 0.02)
    Po, Qo = P0.copy(), Q0.copy()
    E = len(ev_u)
    rp = np.array(epoch_round_ptr(ev_ptr, 3), np.int64)
    nll_o = _epoch_oracle(orc)(Po, Qo, ev_u, ev_i, j, rp, 0.05, 0.01, 0.02)
    P, Q = dev.get_factors()
    assert rel_err(P, Po) < TOL and rel_err(Q, Qo) < TOL and abs(nll - nll_o) <= 1e-9 * max(abs(nll_o), 1e-30)
    dev.set_factors(P0, Q0)
    ok = j >= 0
    nll = dev.bpr_replay(ev_u[ok], ev_i[ok], j[ok], 0.05, 0.01, 0.02)
    Po, Qo = P0.copy(), Q0.copy()
    nll_o = orc.bpr_sequential(Po, Qo, ev_u[ok], ev_i[ok], j[ok], 0.05, 0.01, 0.02)
    P, Q = dev.get_factors()
    assert rel_err(P, Po) < 1e-6 and rel_err(Q, Qo) < 1e-6 and abs(nll - nll_o) <= 1e-9 * max(abs(nll_o), 1e-30)
    if k <= 256:
        mp = np.zeros(m + 1, np.int64)
        users = np.arange(m, dtype=np.int32)
        N = min(n, 3)
        ids, sc = dev.topn_scan(users, N, mp, np.zeros(0, np.int32))
        oid, osc, rc = orc.topn_scan(P, Q, users, N, mp, np.zeros(0, np.int32))
        assert rc == 0 and np.array_equal(ids, oid) and np.array_equal(sc, osc)
        assert np.array_equal(dev.scores(0), orc.scores(P, Q, 0))
    dev.close()


def test_epoch_with_the_single_precision_coefficient_stays_within_tolerance(orc):
    """The epoch path's default on one GPU (k_round_u with option round_fast = 1: the step's coefficient lr (1 - sigmoid(x)) in
    single precision) against the restatement with the reference's double-precision sigmoid: factors within 1e-5 after two
    epochs, the loss (double-precision sigmoids of the device's own margins) within 1e-6."""
    if not _SEQ[0]:
        pytest.skip('k_round_u only')
    from yue_amd._shim import Device
    for (m, n, d, k, W) in [(2000, 3000, 50, 128, 8192), (500, 257, 30, 64, 1000), (300, 400, 20, 200, 128)]:
        data = synth.make_arrays(m, n, d, seed=11)
        ev_u = np.repeat(np.arange(m, dtype=np.int32), np.diff(data['ev_ptr']))
        P0, Q0 = synth.init_factors(m, n, k, 12)
        dev = Device(0, raise_errors=True)
        dev.set_option('round_fast', 1)
        dev.set_factors(P0, Q0)
        dev.set_interactions(data['indptr'], data['indices'], data['ev_ptr'], data['ev_i'])
        rp = np.array(epoch_round_ptr(data['ev_ptr'], W), np.int64)
        Po, Qo = P0.copy(), Q0.copy()
        for epoch in range(2):
            j = dev.sample_negatives(3, epoch)
            nll, _, _ = dev.bpr_epoch(3, epoch, W, 0.02, 0.01, 0.01)
            assert dev.get_option('round_last_user_seq') == 1
            nll_o = orc.bpr_rounds_seq_user(Po, Qo, ev_u, data['ev_i'], j, rp, 0.02, 0.01, 0.01)
            assert abs(nll - nll_o) <= 1e-6 * abs(nll_o)
        P, Q = dev.get_factors()
        dev.close()
        print('k=%d: rel P %.2e Q %.2e, bit-equal P %.3f Q %.3f' % (k, rel_err(P, Po), rel_err(Q, Qo), np.mean(P == Po), np.mean(Q == Qo)))
        assert rel_err(P, Po) < TOL and rel_err(Q, Qo) < TOL


@pytest.mark.parametrize('k', [64, 128, 200])
def test_epoch_with_users_of_more_than_one_header_segment(orc, k):
    """150 events per user: k_round_u walks a user in segments of 64 events (headers in register lanes, the gather ring wraps
    inside a segment); k_round_m takes them 8 at a time anyway.  Against the restatement of the module's current semantics."""
    from yue_amd._shim import Device
    m, n, d, W = 300, 2000, 150, 4000
    data, P0, Q0, ev_u = _synth_problem(m, n, d, k, seed=31)
    rp = np.array(epoch_round_ptr(data['ev_ptr'], W), np.int64)
    dev = Device(0, raise_errors=True)
    dev.set_factors(P0, Q0)
    dev.set_interactions(data['indptr'], data['indices'], data['ev_ptr'], data['ev_i'])
    Po, Qo = P0.copy(), Q0.copy()
    for epoch in range(2):
        j = dev.sample_negatives(8, epoch)
        nll, _, _ = dev.bpr_epoch(8, epoch, W, 0.02, 0.01, 0.01)
        nll_o = _epoch_oracle(orc)(Po, Qo, ev_u, data['ev_i'], j, rp, 0.02, 0.01, 0.01)
        assert abs(nll - nll_o) <= 1e-9 * abs(nll_o)
    P, Q = dev.get_factors()
    dev.close()
    assert rel_err(P, Po) < TOL and rel_err(Q, Qo) < TOL
