"""Seeded synthetic implicit-feedback logs (ours; the reference ships no dataset).

Shape follows SURVEY.md section 8(d): every user has ``d`` events, the item of
event ``(user, slot)`` is ``floor(n * r**2)`` with ``r`` a counter-based uniform
in [0,1) (popularity skewed towards low item ids), duplicates are kept in the
event list and removed only in the sorted-unique CSR used for negative rejection.

Two outputs:
  * ``write_text_log``  -> ``time,user,track,artist`` text for the config-driven
    pipeline (BPR.conf, C1);
  * ``make_arrays``     -> array-native events + CSR for C2..C5 (H5 in SURVEY.md).
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    """Vectorised splitmix64 finaliser on uint64 arrays."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over='ignore'):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        x = x ^ (x >> np.uint64(31))
    return x


def event_items(seed, users, slots, n):
    """Item id of event (user, slot): floor(n * r^2), r = 53-bit uniform."""
    users = np.asarray(users, dtype=np.uint64)
    slots = np.asarray(slots, dtype=np.uint64)
    with np.errstate(over='ignore'):
        key = (np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15)
               + users * np.uint64(0xD1B54A32D192ED03)
               + slots * np.uint64(0x8CB92BA72F3D8DD7)) & _M64
    r = (splitmix64(key) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    it = np.floor(n * r * r).astype(np.int64)
    return np.minimum(it, n - 1)


def make_arrays(m, n, d, seed=20260001, user_chunk=1 << 18):
    """Array-native training interactions for m users x n items, d events/user.

    Returns dict with
      ev_ptr  int64[m+1]  events of user u are ev_i[ev_ptr[u]:ev_ptr[u+1]] (slot order)
      ev_i    int32[E]    positive item of every event (duplicates kept)
      indptr  int64[m+1], indices int32[nnz]  sorted-unique items per user
    """
    ev_i = np.empty(m * d, dtype=np.int32)
    uniq_rows = []
    counts = np.empty(m, dtype=np.int64)
    slots = np.arange(d, dtype=np.uint64)[None, :]
    for u0 in range(0, m, user_chunk):
        u1 = min(m, u0 + user_chunk)
        users = np.arange(u0, u1, dtype=np.uint64)[:, None]
        it = event_items(seed, users, slots, n).astype(np.int32)      # [u, d]
        ev_i[u0 * d:u1 * d] = it.reshape(-1)
        srt = np.sort(it, axis=1)
        keep = np.ones(srt.shape, dtype=bool)
        keep[:, 1:] = srt[:, 1:] != srt[:, :-1]
        counts[u0:u1] = keep.sum(axis=1)
        uniq_rows.append(srt[keep])
    indptr = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(counts, out=indptr[1:])
    indices = np.concatenate(uniq_rows).astype(np.int32)
    ev_ptr = np.arange(m + 1, dtype=np.int64) * d
    return {'m': m, 'n': n, 'ev_ptr': ev_ptr, 'ev_i': ev_i,
            'indptr': indptr, 'indices': indices}


def text_events(m, n, d, seed=20260001):
    """Events of the C1-style text log in file order (slot-major, user-minor)."""
    rows = []
    for s in range(d):
        it = event_items(seed, np.arange(m), np.full(m, s), n)
        for u in range(m):
            i = int(it[u])
            rows.append(('%010d' % s, 'u%d' % u, 't%d' % i, 'a%d' % (i % 50)))
    return rows


def write_text_log(path, m, n, d, seed=20260001):
    """``time,user,track,artist`` lines, readable with BPR.conf's record.setup."""
    with open(path, 'w') as f:
        for t, u, i, a in text_events(m, n, d, seed):
            f.write('%s,%s,%s,%s\n' % (t, u, i, a))


def init_factors(m, n, k, seed=20260002):
    """Same draw order as IterativeRecommender.initModel (P first, then Q)."""
    rs = np.random.RandomState(seed)
    P = rs.rand(m, k).astype(np.float32) / 10
    Q = rs.rand(n, k).astype(np.float32) / 10
    return P, Q


def make_test_arrays(m, n, d, d_test, indptr, indices, seed=20260001, user_chunk=1 << 18):
    """Held-out items per user for the array-native path: slots d .. d+d_test-1 of the same stream,
    de-duplicated and with the user's training items removed.  Returns (test_indptr, test_indices)."""
    rows, counts = [], np.empty(m, dtype=np.int64)
    slots = np.arange(d, d + d_test, dtype=np.uint64)[None, :]
    for u0 in range(0, m, user_chunk):
        u1 = min(m, u0 + user_chunk)
        users = np.arange(u0, u1, dtype=np.uint64)[:, None]
        it = np.sort(event_items(seed, users, slots, n), axis=1)
        keep = np.ones(it.shape, dtype=bool)
        keep[:, 1:] = it[:, 1:] != it[:, :-1]
        ukey = np.repeat(np.arange(u0, u1, dtype=np.int64), d_test).reshape(it.shape) * n + it
        train_keys = np.repeat(np.arange(u0, u1, dtype=np.int64), np.diff(indptr[u0:u1 + 1])) * n + indices[indptr[u0]:indptr[u1]]
        keep &= ~np.isin(ukey, train_keys)
        counts[u0:u1] = keep.sum(axis=1)
        rows.append(it[keep])
    tp = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(counts, out=tp[1:])
    return tp, np.concatenate(rows).astype(np.int32)


def write_csr(path, m, n, d, d_test=6, seed=20260001):
    """The synthetic problem as a binary csr data set (``record.setup=-format csr``): d training events and up to
    d_test held-out items per user."""
    from .data.arrays import save_csr
    data = make_arrays(m, n, d, seed=seed)
    tp, ti = make_test_arrays(m, n, d, d_test, data['indptr'], data['indices'], seed=seed)
    save_csr(path, m, n, data['ev_ptr'], data['ev_i'], tp, ti)
    return data, tp, ti


if __name__ == '__main__':
    # python -m yue_amd.synth [path] [users] [items] [events per user]: the seeded text log the shipped config/*.conf read
    import os
    import sys
    path = sys.argv[1] if len(sys.argv) > 1 else './dataset/log.txt'
    m, n, d = (int(x) for x in (sys.argv[2:5] + ['1000', '1000', '20'][len(sys.argv[2:5]):]))
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    write_text_log(path, m, n, d)
    print('wrote %s: %d users x %d items, %d events per user' % (path, m, n, d))
