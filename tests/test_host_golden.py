"""CPU: the host-side mirror of the reference interface (config, file, Record, Measure) against
golden outputs of the reference itself (tools/make_goldens.py)."""
import os

import numpy as np

from yue_amd import synth
from yue_amd.data.record import Record
from yue_amd.evaluation.measure import Measure
from yue_amd.tool.config import Config, LineConfig
from yue_amd.tool.file import FileIO
from util import csr_from_events, gj, gz

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _conf_text(overrides=None, extra=None):
    """key=value lines of the reference's BPR.conf as captured in the golden (data, not a copy of the file)."""
    kv = dict(gj('g1_config.json')['bpr_conf'])
    kv.update(overrides or {})
    kv.update(extra or {})
    return '\n'.join('%s=%s' % item for item in kv.items()) + '\n'


def test_config_matches_reference(tmp_path):
    g = gj('g1_config.json')
    path = tmp_path / 'BPR.conf'
    path.write_text(_conf_text(extra={'bpr.hip': '-mode replay -gpu 0'}) + 'a malformed line\n\n')
    conf = Config(str(path))
    ours = dict(conf.config)
    assert ours.pop('bpr.hip') == '-mode replay -gpu 0'     # our optional key; the rest parses as in the reference
    assert ours == g['bpr_conf']
    for case in g['lineconfig']:
        lc = LineConfig(case['line'])
        assert lc.isMainOn() == case['main'], case['line']
        assert lc.options == case['options'], case['line']
    assert conf.contains('record') and not conf.contains('nope')


def _c1_conf(tmp_path, k=10, iters=1, topn='5,10', extra=''):
    log = tmp_path / 'log.txt'
    if not log.exists():
        synth.write_text_log(str(log), 1000, 1000, 20)
    text = _conf_text({'record': str(log), 'num.factors': str(k), 'num.max.iter': str(iters), 'item.ranking': '-topN ' + topn,
                       'output.setup': 'on -dir ' + str(tmp_path / 'results') + '/'}, {'bpr.hip': '-mode replay -gpu 0'})
    p = tmp_path / ('c1_%d_%d.conf' % (k, iters))
    p.write_text(text + extra)
    return Config(str(p))


def _load(conf):
    setup = LineConfig(conf['record.setup'])
    cols = {}
    for col in setup['-columns'].split(','):
        a, b = col.split(':')
        cols[a] = int(b)
    return FileIO.loadDataSet(conf['record'], columns=cols, delim=setup['-delim'])


def test_record_matches_reference(tmp_path):
    g = gj('g2_record_c1.json')
    ev = gz('g2_events_c1.npz')
    conf = _c1_conf(tmp_path)
    rec = Record(conf, _load(conf), [])
    assert rec.getSize('user') == g['m'] and rec.getSize('track') == g['n']
    assert [rec.id2name['user'][i] for i in range(g['m'])] == g['users']
    assert [rec.id2name['track'][i] for i in range(g['n'])] == g['items']
    assert len(rec.trainingData) == g['len_trainingData'] and rec.recordCount == g['recordCount']
    assert [[u, [list(kv) for kv in rec.testSet[u].items()]] for u in rec.testSet] == g['testSet']
    arr = rec.to_arrays('track')
    ev_u = np.repeat(np.arange(g['m'], dtype=np.int32), np.diff(arr['ev_ptr']))
    assert np.array_equal(ev_u, ev['ev_u']) and np.array_equal(arr['ev_i'], ev['ev_i'])
    indptr, indices = csr_from_events(ev['ev_u'], ev['ev_i'], g['m'])
    assert np.array_equal(arr['indptr'], indptr) and np.array_equal(arr['indices'], indices)


def test_measure_matches_reference():
    g = gj('g7_measure.json')
    assert Measure.rankingMeasure(g['origin'], g['res'], g['N'], g['itemCount']) == g['measure']
