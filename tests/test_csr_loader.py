"""The binary integer data set behind ``record.setup=-format csr`` (yue_amd/data/arrays.py: save_csr / load_csr) and its
route through the driver (yue_amd/yue.py) -- host logic, no GPU.  SURVEY 8(f) row 2; stands where the reference has
tool/file.py:23-52 + data/record.py:138-226 for text logs."""
import numpy as np
import pytest

from yue_amd import synth
from yue_amd.data.arrays import ArrayRecord, load_csr, save_csr
from yue_amd.tool.config import Config
from test_host_golden import _conf_text


def test_round_trip_and_record_surface(tmp_path):
    path = str(tmp_path / 'small.npz')
    data, tp, ti = synth.write_csr(path, 300, 200, 12, d_test=4, seed=3)
    rec = load_csr(path)
    assert isinstance(rec, ArrayRecord) and rec.getSize('user') == 300 and rec.getSize('track') == 200
    assert np.array_equal(rec.ev_ptr, data['ev_ptr']) and np.array_equal(rec.ev_i, data['ev_i'])
    assert np.array_equal(rec.indptr, data['indptr']) and np.array_equal(rec.indices, data['indices'])
    assert np.array_equal(rec.test_indptr, tp) and np.array_equal(rec.test_indices, ti)
    u = int(np.flatnonzero(np.diff(tp) > 0)[0])
    assert rec.testSet[str(u)] == {str(int(i)): 1 for i in ti[tp[u]:tp[u + 1]]}
    assert [e['track'] for e in rec.userRecord[str(u)]] == [str(int(i)) for i in data['ev_i'][data['ev_ptr'][u]:data['ev_ptr'][u + 1]]]
    assert rec.getId('7', 'track') == 7 and len(rec.trainingData) == 300 * 12
    # no test items: still a valid data set
    save_csr(str(tmp_path / 'notest.npz'), 3, 5, [0, 1, 1, 3], [4, 0, 2])
    assert len(load_csr(str(tmp_path / 'notest.npz')).testSet) == 0


@pytest.mark.parametrize('breakage', ['missing', 'offsets', 'range', 'not_npz'])
def test_malformed_files_exit_like_the_reference(tmp_path, capsys, breakage):
    path = str(tmp_path / 'bad.npz')
    if breakage == 'missing':
        np.savez(path, m=np.int64(2), n=np.int64(3), ev_ptr=np.array([0, 1, 2]))
    elif breakage == 'offsets':
        save_csr(path, 2, 3, [0, 1, 5], [0, 1])
    elif breakage == 'range':
        save_csr(path, 2, 3, [0, 1, 2], [0, 3])
    else:
        open(path, 'w').write('time,user,track\n')
    with pytest.raises(SystemExit) as stop:
        load_csr(path)
    assert stop.value.code == -1 and 'csr data set' in capsys.readouterr().out


def test_driver_takes_the_csr_route(tmp_path, capsys):
    from yue_amd.yue import Yue
    path = str(tmp_path / 'small.npz')
    synth.write_csr(path, 50, 40, 6, d_test=2, seed=1)
    text = _conf_text({'record': path, 'record.setup': '-format csr', 'evaluation.setup': '-target track',
                       'output.setup': 'on -dir ' + str(tmp_path / 'results') + '/'}, {'bpr.hip': '-mode epoch -gpu 0'})
    conf_file = tmp_path / 'csr.conf'
    conf_file.write_text(text)
    y = Yue(Config(str(conf_file)))
    assert isinstance(y.trainingData, ArrayRecord) and y.trainingData.m == 50 and y.testData == []
    conf_file.write_text(text.replace('evaluation.setup=-target track', 'evaluation.setup=-target track -cv 3'))
    with pytest.raises(SystemExit):
        Yue(Config(str(conf_file)))
    assert '-cv needs the text log' in capsys.readouterr().out
