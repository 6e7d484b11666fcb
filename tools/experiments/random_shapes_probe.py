import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import oracle
from yue_amd._shim import Device
from yue_amd.dist import epoch_round_ptr
from util import rel_err
orc = oracle.Oracle()
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 10)
worst = 0.0
for case in range(lo, hi):
    rs = np.random.RandomState(1000 + case)
    m = int(rs.randint(1, 1500)); n = int(rs.randint(2, 4000)); k = int(rs.choice([3, 7, 16, 33, 64, 100, 128, 200]))
    W = int(rs.choice([1, 5, 37, 256, 1000, 4096, 100000]))
    P0 = rs.rand(m, k).astype(np.float32) / 10
    Q0 = rs.rand(n, k).astype(np.float32) / 10
    pop = rs.zipf(1.3, size=n).astype(np.float64); pop /= pop.sum()
    ev, rows = [], []
    for u in range(m):
        cnt = int(rs.randint(0, min(60, n - 1) + 1)) if rs.rand() > 0.1 else 0
        it = rs.choice(n, size=cnt, p=pop) if cnt else np.zeros(0, np.int64)
        ev.append(it.astype(np.int32)); rows.append(np.unique(it).astype(np.int32))
    ev_ptr = np.cumsum([0] + [len(e) for e in ev]).astype(np.int64)
    if ev_ptr[-1] == 0: continue
    ev_i = np.concatenate(ev).astype(np.int32)
    indptr = np.cumsum([0] + [len(r) for r in rows]).astype(np.int64)
    indices = np.concatenate(rows).astype(np.int32)
    ev_u = np.repeat(np.arange(m, dtype=np.int32), np.diff(ev_ptr))
    rp = np.array(epoch_round_ptr(ev_ptr, W), np.int64)
    top = np.bincount(ev_i, minlength=n).max()
    for meta in (1, 0):
        dev = Device(0, raise_errors=True)
        dev.set_option('round_meta', meta)
        stage = int(rs.choice([1, 1, 0, 2, 5, 16, 64]))
        dev.set_option('round_stage', stage)
        dev.set_factors(P0, Q0)
        dev.set_interactions(indptr, indices, ev_ptr, ev_i)
        Po, Qo = P0.copy(), Q0.copy()
        out = []
        for epoch in range(2):
            j = orc.sample_counter(77, epoch, ev_u, n, indptr, indices)
            nll, sp, sq = dev.bpr_epoch(77, epoch, W, 0.03, 0.01, 0.02)
            nll_o = orc.bpr_rounds(Po, Qo, ev_u, ev_i, j, rp, 0.03, 0.01, 0.02)
            P, Q = dev.get_factors()
            out.append((rel_err(P, Po), rel_err(Q, Qo), abs(nll - nll_o) / max(abs(nll_o), 1e-30)))
        dev.close()
        bad = max(max(o[0], o[1]) for o in out) >= 1e-5 or max(o[2] for o in out) > 1e-9
        worst = max(worst, max(max(o[0], o[1]) for o in out))
        if bad or hi - lo <= 10:
            print('%scase %d m %d n %d k %d W %d E %d hottest %d rounds %d meta %d stage %d:' % ('FAIL ' if bad else '', case, m, n, k, W, len(ev_i), top, len(rp) - 1, meta, stage), ' '.join('P %.1e Q %.1e nll %.1e' % o for o in out), flush=True)
print('cases %d..%d done, worst rel err %.2e' % (lo, hi, worst))
