"""GPU parity of CUNE's two-level BPR loop, through the C ABI: yue_cune_steps against the reference's
own outputs (tests/golden/g9_*) and the NumPy oracle that is bit-equal to them.  Tolerances: the reference's BLAS dot
order is not pinned (ours is the 64-lane butterfly) -> factors within 1e-5 rel (BASELINE.json north_star), per-step
losses within 1e-5."""
import numpy as np
import pytest

from util import gj, gz, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('tag', ['cune_d2_k20_e2', 'cune_d3_k64_e1'])
def test_steps_match_reference_and_oracle(orc, tag):
    from oracle.numpy_cune import cune_step
    from yue_amd._shim import Device
    z, meta = gz('g9_%s.npz' % tag), gj('g9_%s.json' % tag)
    iters, s = int(z['iters']), float(z['s'])
    per = len(z['u']) // iters
    sl = slice(0, per)                                   # the first epoch runs at the initial learning rate
    dev = Device(0, raise_errors=True)
    dev.set_factors(z['P0'], z['Q0'])
    loss = dev.cune_steps(z['u'][sl], z['i'][sl], z['kk'][sl], z['j'][sl], s, meta['lr0'], meta['regU'], meta['regI'])
    P, Q = dev.get_factors()
    Po, Qo = z['P0'].copy(), z['Q0'].copy()
    loss_o = np.array([cune_step(Po, Qo, int(z['u'][t]), int(z['i'][t]), int(z['kk'][t]), int(z['j'][t]), s, meta['lr0'], meta['regU'], meta['regI'])
                       for t in range(per)])
    assert rel_err(P, Po) < 1e-5 and rel_err(Q, Qo) < 1e-5
    assert np.abs(loss - loss_o).max() < 1e-5 * np.abs(loss_o).max()
    assert (z['kk'][sl] == z['j'][sl]).any()             # steps whose friends' item is also their negative (one row, two names)
    # the whole run: loss as the reference accumulates it (float32 once the per-user regulariser joins, :175) and its printed line
    if iters == 1:
        assert rel_err(P, z['P']) < 1e-5 and rel_err(Q, z['Q']) < 1e-5
    dev.set_factors(z['P0'], z['Q0'])
    total, last_u, start = 0, None, 0
    u = z['u'][sl]
    bounds = [0] + [t + 1 for t in range(per - 1) if u[t + 1] != u[t]] + [per]
    for a, b in zip(bounds, bounds[1:]):                 # one device call per user, then the regulariser as the reference adds it
        for x in dev.cune_steps(z['u'][a:b], z['i'][a:b], z['kk'][a:b], z['j'][a:b], s, meta['lr0'], meta['regU'], meta['regI']):
            total += float(x)
        sp, sq = dev.sumsq()
        total += meta['regU'] * np.float32(sp) + meta['regI'] * np.float32(sq)
    first_line_loss = float(meta['lines'][0].split('loss = ')[1].split(',')[0])
    assert abs(float(total) - first_line_loss) < 2e-6 * first_line_loss and type(total).__name__ == 'float32'
    dev.close()


def test_steps_reject_bad_input():
    from yue_amd._shim import Device, YueHipError
    dev = Device(0, raise_errors=True)
    rs = np.random.RandomState(0)
    dev.set_factors(rs.rand(4, 8).astype(np.float32), rs.rand(6, 8).astype(np.float32))
    one = lambda v: np.array([v], np.int32)
    with pytest.raises(YueHipError):
        dev.cune_steps(one(0), one(1), one(1), one(2), 2.0, 0.02, 0.01, 0.01)      # k == i
    with pytest.raises(YueHipError):
        dev.cune_steps(one(0), one(7), one(1), one(2), 2.0, 0.02, 0.01, 0.01)      # item out of range
    assert len(dev.cune_steps(one(0)[:0], one(0)[:0], one(0)[:0], one(0)[:0], 2.0, 0.02, 0.01, 0.01)) == 0
    dev.close()


def test_through_the_plugin_surface(tmp_path, capsys):
    # CUNE.conf keys on the d2 log, the friends' item sets of the golden run injected, driven as tools/make_goldens.py drives
    # the reference's loop: same seeds -> same draws -> the reference's printed lines and factors
    import random
    from yue_amd import synth
    from yue_amd.recommender.advanced.CUNE import CUNE
    from yue_amd.tool.config import Config
    from test_host_golden import _conf_text, _load
    tag = 'cune_d2_k20_e2'
    z, meta = gz('g9_%s.npz' % tag), gj('g9_%s.json' % tag)
    m, n, d = meta['dataset']
    log = tmp_path / 'log.txt'
    synth.write_text_log(str(log), m, n, d)
    text = _conf_text({'record': str(log), 'recommender': 'CUNE', 'num.factors': str(int(z['k'])), 'num.max.iter': str(int(z['iters'])),
                       'learnRate': '-init 0.02 -max 0.1', 'reg.lambda': '-u 0.01 -i 0.01 -b 0.01 -s 0.2',
                       'output.setup': 'on -dir ' + str(tmp_path / 'results') + '/'}, {'CUNE': '-T 20 -L 10 -l 20 -w 5 -k 50 -s 2 -ep 10'})
    path = tmp_path / 'cune.conf'
    path.write_text(text)
    conf = Config(str(path))
    rec = CUNE(conf, _load(conf), [])
    rec.readConfiguration()
    seed = int(z['seed'])
    random.seed(seed)
    np.random.seed(seed)
    rec.initModel()
    assert np.array_equal(rec.P, z['P0']) and np.array_equal(rec.Q, z['Q0'])
    names = rec.data.id2name[rec.recType]
    unames = rec.data.id2name['user']
    from collections import defaultdict
    rec.IPositiveSet = defaultdict(list)
    for u in range(int(z['m'])):
        rec.IPositiveSet[unames[u]] = [names[int(x)] for x in z['ip_items'][z['ip_ptr'][u]:z['ip_ptr'][u + 1]]]
    random.seed(seed + 2)
    capsys.readouterr()
    rec.buildModel()
    lines = [ln for ln in capsys.readouterr().out.splitlines() if 'iteration' in ln]
    assert len(lines) == len(meta['lines'])
    for got, want in zip(lines, meta['lines']):          # float32 losses of ~1e4: the printed 4th decimal is below float32 resolution
        g, w = float(got.split('loss = ')[1].split(',')[0]), float(want.split('loss = ')[1].split(',')[0])
        assert abs(g - w) < 2e-6 * w and got.split('learning_Rate')[1] == want.split('learning_Rate')[1]
    assert rel_err(rec.P, z['P']) < 1e-5 and rel_err(rec.Q, z['Q']) < 1e-5
    assert abs(rec.lRate - float(z['lRate'])) < 1e-12 and type(rec.loss).__name__ == 'float32'
