"""Array-native stand-in for ``Record``.

The reference's data model is dict-of-dict-of-str built from a text log; 1M users x 200K items do
not fit through it.  ``ArrayRecord`` offers the part of the ``Record`` surface the BPR path reads
(``getSize / getId / id2name / name2id / testSet / userRecord / to_arrays``) on top of integer
arrays; object names are the decimal strings of the ids.  Pass it as ``trainingSet`` to a
recommender (``BPR(conf, ArrayRecord(...))``); only the throughput mode (``bpr.hip=-mode epoch``)
can train on it -- the replay mode draws negatives through Python's ``random.choice`` over item
NAMES, which is exactly what does not scale.
"""
import numpy as np


class _Names(object):
    """id -> name and name -> id for names that are str(id); behaves like the dicts it replaces."""

    def __init__(self, count, to_name):
        self._count = count
        self._to_name = to_name

    def __len__(self):
        return self._count

    def __contains__(self, key):
        try:
            return 0 <= int(key) < self._count
        except (TypeError, ValueError):
            return False

    def __getitem__(self, key):
        if key not in self:
            raise KeyError(key)
        return str(int(key)) if self._to_name else int(key)

    def __iter__(self):
        return (str(i) if not self._to_name else i for i in range(self._count))

    def keys(self):
        return iter(self)


class _CsrDictView(object):
    """user name -> {item name: 1} (testSet) or -> [{recType: item name}, ...] (userRecord), built on demand."""

    def __init__(self, indptr, indices, rec_type, as_events):
        self._indptr, self._indices, self._rec_type, self._as_events = indptr, indices, rec_type, as_events
        self._rows = np.flatnonzero(np.diff(indptr) > 0)

    def __len__(self):
        return len(self._rows)

    def __iter__(self):
        return (str(int(u)) for u in self._rows)

    def keys(self):
        return iter(self)

    def __contains__(self, user):
        try:
            u = int(user)
        except (TypeError, ValueError):
            return False
        return 0 <= u < len(self._indptr) - 1 and self._indptr[u + 1] > self._indptr[u]

    def __getitem__(self, user):
        u = int(user)
        row = self._indices[self._indptr[u]:self._indptr[u + 1]]
        if self._as_events:
            return [{self._rec_type: str(int(i))} for i in row]
        return {str(int(i)): 1 for i in row}

    def user_ids(self):
        return self._rows


class _Sized(object):
    def __init__(self, n):
        self._n = n

    def __len__(self):
        return self._n


class ArrayRecord(object):
    """Integer-array data set.

    ev_ptr[m+1], ev_i[E]            training events, user-major (users by ascending id), duplicates kept
    test_indptr[m+1], test_indices  test items per user (sorted unique, disjoint from the training items)
    """

    def __init__(self, m, n, ev_ptr, ev_i, test_indptr=None, test_indices=None, rec_type='track'):
        self.m, self.n, self.recType = int(m), int(n), rec_type
        self.ev_ptr = np.ascontiguousarray(ev_ptr, np.int64)
        self.ev_i = np.ascontiguousarray(ev_i, np.int32)
        assert len(self.ev_ptr) == self.m + 1 and self.ev_ptr[-1] == len(self.ev_i)
        ev_u = np.repeat(np.arange(self.m, dtype=np.int64), np.diff(self.ev_ptr))
        keys = np.unique(ev_u * self.n + self.ev_i)
        indptr = np.zeros(self.m + 1, np.int64)
        np.add.at(indptr, keys // self.n + 1, 1)
        self.indptr = np.cumsum(indptr)
        self.indices = (keys % self.n).astype(np.int32)
        if test_indptr is None:
            test_indptr, test_indices = np.zeros(self.m + 1, np.int64), np.zeros(0, np.int32)
        self.test_indptr = np.ascontiguousarray(test_indptr, np.int64)
        self.test_indices = np.ascontiguousarray(test_indices, np.int32)
        self.id2name = {'user': _Names(self.m, True), rec_type: _Names(self.n, True)}
        self.name2id = {'user': _Names(self.m, False), rec_type: _Names(self.n, False)}
        self.testSet = _CsrDictView(self.test_indptr, self.test_indices, rec_type, False)
        self.userRecord = _CsrDictView(self.ev_ptr, self.ev_i, rec_type, True)
        self.trackRecord = {}
        self.PopTrack = {}
        self.trainingData = _Sized(len(self.ev_i))
        self.recordCount = len(self.ev_i)

    def getSize(self, t):
        return len(self.name2id[t])

    def getId(self, obj, t):
        if obj in self.name2id[t]:
            return int(obj)
        print('No ' + t + ' ' + str(obj) + ' exists!')
        exit(-1)

    def contains(self, obj, t):
        return obj in self.name2id[t]

    def printTrainingSize(self):
        print('user count:', self.m)
        print(self.recType + ' count:', self.n)
        print('Training set size:', self.recordCount)

    def to_arrays(self, recType):
        assert recType == self.recType
        return {'ev_ptr': self.ev_ptr, 'ev_i': self.ev_i, 'indptr': self.indptr, 'indices': self.indices}


CSR_KEYS = ('m', 'n', 'ev_ptr', 'ev_i', 'test_indptr', 'test_indices')


def save_csr(path, m, n, ev_ptr, ev_i, test_indptr=None, test_indices=None):
    """Binary data set for ``record.setup=-format csr`` (an uncompressed .npz of plain integer arrays):
    m, n, ev_ptr int64[m+1], ev_i int32[E] (training events, user-major), test_indptr int64[m+1],
    test_indices int32 (held-out items per user, sorted unique)."""
    if test_indptr is None:
        test_indptr, test_indices = np.zeros(int(m) + 1, np.int64), np.zeros(0, np.int32)
    np.savez(path, m=np.int64(m), n=np.int64(n), ev_ptr=np.ascontiguousarray(ev_ptr, np.int64), ev_i=np.ascontiguousarray(ev_i, np.int32),
             test_indptr=np.ascontiguousarray(test_indptr, np.int64), test_indices=np.ascontiguousarray(test_indices, np.int32))


def load_csr(path, rec_type='track'):
    """The loader behind ``record.setup=-format csr``: the counterpart of FileIO.loadDataSet + Record's id
    assignment (reference tool/file.py:23-52, data/record.py:138-226) for data that is already integer ids.
    Nothing in the file is executed (allow_pickle stays off); malformed files end in the reference's
    print-and-exit convention."""
    try:
        with np.load(path, allow_pickle=False) as z:
            missing = [key for key in CSR_KEYS if key not in z.files]
            if missing:
                print('The csr data set %s lacks the arrays: %s' % (path, ', '.join(missing)))
                exit(-1)
            arrays = {key: z[key] for key in CSR_KEYS}
    except (OSError, ValueError) as err:
        print('Cannot read the csr data set %s: %s' % (path, err))
        exit(-1)
    m, n = int(arrays['m']), int(arrays['n'])
    ev_ptr, ev_i, tp, ti = arrays['ev_ptr'], arrays['ev_i'], arrays['test_indptr'], arrays['test_indices']
    ok = (m > 0 and n > 0 and len(ev_ptr) == m + 1 and len(tp) == m + 1 and ev_ptr[0] == 0 and tp[0] == 0
          and ev_ptr[-1] == len(ev_i) and tp[-1] == len(ti) and (np.diff(ev_ptr) >= 0).all() and (np.diff(tp) >= 0).all()
          and (len(ev_i) == 0 or (0 <= ev_i.min() and ev_i.max() < n)) and (len(ti) == 0 or (0 <= ti.min() and ti.max() < n)))
    if not ok:
        print('The csr data set %s is inconsistent (sizes, offsets or ids out of range).' % path)
        exit(-1)
    return ArrayRecord(m, n, ev_ptr, ev_i, tp, ti, rec_type)


def ranking_measure_ids(test_indptr, test_indices, users, ids, top, item_count):
    """evaluation/measure.py:16-66,91-101 on integer lists: ``ids[nu, N]`` are the lists of ``users``
    (int ids).  Same definitions and output format as Measure.rankingMeasure, vectorised with NumPy
    (sums are taken in a different order, so the last printed digits can differ)."""
    users = np.asarray(users, np.int64)
    nu = len(users)
    tlen = (test_indptr[users + 1] - test_indptr[users]).astype(np.float64)
    span = int(max(int(test_indices.max()) if len(test_indices) else 0, int(ids.max())) + 1)
    tkeys = np.repeat(np.arange(len(test_indptr) - 1, dtype=np.int64), np.diff(test_indptr)) * span + test_indices
    out = []
    for n in top:
        cut = ids[:, :n].astype(np.int64)
        hit = np.isin(users[:, None] * span + cut, tkeys)                 # [nu, n] bool
        # hits = |set(test) & set(list)|: a duplicated listed item counts once
        srt = np.sort(np.where(hit, cut, -1 - np.arange(n)[None, :]), axis=1)
        dup = np.zeros_like(hit)
        dup[:, 1:] = srt[:, 1:] == srt[:, :-1]
        hits = hit.sum(axis=1) - (dup & (srt >= 0)).sum(axis=1)
        prec = float(hits.sum()) / (nu * n)
        rec = float((hits / tlen).sum() / nu)
        f1 = 2 * prec * rec / (prec + rec) if (prec + rec) != 0 else 0
        # MAP counts every listed hit position (duplicates included), as the reference does
        csum = np.cumsum(hit, axis=1)
        ap = (np.where(hit, csum / (np.arange(n)[None, :] + 1.0), 0.0).sum(axis=1) / np.minimum(tlen, n)).sum() / nu
        cov = len(np.unique(cut)) / float(item_count)
        out += ['Top ' + str(n) + '\n', 'Precision:' + str(prec) + '\n', 'Recall:' + str(rec) + '\n', 'F1:' + str(f1) + '\n',
                'MAP:' + str(float(ap)) + '\n', 'Coverage:' + str(cov) + '\n']
    return out
