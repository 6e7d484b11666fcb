"""NumPy restatement of the reference's FISM (recommender/cf/FISM.py) -- TEST INFRASTRUCTURE.

SURVEY 8(f) rank 3.  Types as the reference leaves them: item-history factors P float64 [n,k], item
factors Q float32 [n,k], item bias Bi float64 [n] (FISM.py:15-18 on top of IterativeRecommender.initModel).
Every operation is the NumPy operation the reference issues (dot through BLAS, python-float coefficients
under NEP-50 promotion, in-place float32 row updates), so on this NumPy build the results equal the
reference's bit for bit (tests/test_oracle_golden.py pins them to tests/golden/g8_*).
"""
import numpy as np


def user_coefficient(n_events, alpha):
    # FISM.py:42, python floats
    return pow(n_events - 1, -alpha)


def fism_epoch(P, Q, Bi, user_ptr, ev_i, negs, rho, alpha, lRate, regI, regB):
    """One pass of FISM.py:38-69 over the users in order.  `negs` holds the accepted negatives in
    processing order (rho per event, users with a single event contribute none).  In place on P, Q, Bi.
    Returns the sum of 0.5*error**2 (the data part of the loss, :58)."""
    assert P.dtype == np.float64 and Q.dtype == np.float32 and Bi.dtype == np.float64
    half_sq = 0
    cursor = 0
    k = P.shape[1]
    for u in range(len(user_ptr) - 1):
        items = ev_i[user_ptr[u]:user_ptr[u + 1]].tolist()
        if len(items) == 1:
            continue                                             # :40-41
        coef = user_coefficient(len(items), alpha)
        hist = np.zeros(k)
        for it in items:                                         # :44-46
            hist += P[it]
        xs = []
        for i in items:
            x = np.zeros(k)
            for _ in range(rho):
                j = int(negs[cursor])
                cursor += 1
                r_pos = coef * (hist - P[i]).dot(Q[i]) + Bi[i]   # :54
                r_neg = coef * (hist - P[j]).dot(Q[j]) + Bi[j]   # :55
                err = 1 - (r_pos - r_neg)
                half_sq += 0.5 * err ** 2
                Bi[i] += lRate * (err - regB * Bi[i])            # :59-60
                Bi[j] -= lRate * (err + regB * Bi[j])
                Q[i] += lRate * (err * coef * (hist - P[i]) - regI * Q[i])     # :61-62, float64 rhs into float32 rows
                Q[j] -= lRate * (err * coef * (hist - P[j]) + regI * Q[j])
                x += err * (Q[i] - Q[j])                         # :63, with the updated rows
            xs.append(x)
        for x, it in zip(xs, items):                             # :66-68
            P[it] += lRate * (1 / float(rho) * coef * x - regI * P[it])
    assert cursor == len(negs)
    return half_sq


def fism_regulariser(P, Q, Bi, regU, regI, regB):
    # FISM.py:70
    return regU * (P * P).sum() + regI * (Q * Q).sum() + regB * (Bi.dot(Bi))


def fism_scores(P, Q, Bi, items):
    """predict (FISM.py:75-83) for a user whose training events are `items` (duplicates count)."""
    hist = np.zeros(P.shape[1])
    for it in items:
        hist += P[int(it)]
    return Bi + Q.dot(hist) - (P * Q).sum(axis=1)


def overwrite_scan(scores, masked, N):
    """The reference's selection (base/IterativeRecommender.py:98-145) on one score vector of any float type:
    candidates = ids ascending minus `masked`; seed with the first N, stable sort descending; then every
    candidate (the first N again) whose score beats the last slot overwrites the first slot strictly below it."""
    masked = set(int(x) for x in masked)
    cand = [t for t in range(len(scores)) if t not in masked]
    seed = sorted(cand[:N], key=lambda t: scores[t], reverse=True)
    if len(seed) < N:
        raise IndexError('fewer than N candidates')
    vals, ids = [scores[t] for t in seed], list(seed)
    for t in cand:
        s = scores[t]
        if vals[N - 1] < s:
            p = 0
            while vals[p] >= s:
                p += 1
            vals[p], ids[p] = s, t
    return ids, vals


def fism_rounds(P, Q, Bi, user_ptr, ev_i, negs, rho, alpha, lRate, regI, regB, round_users):
    """Round semantics of the throughput path (ours): the users are cut into consecutive rounds of
    `round_users` users.  Inside a round every user runs the reference's whole per-user loop (FISM.py:39-68, draws and
    item-history updates in order) on the model AS IT WAS WHEN THE ROUND STARTED plus the user's own changes so far;
    per row the differences (user's final row - round-start row) are summed over the round's users and added once.
    Rounds are applied in order.  One user per round is the reference loop up to x + (x' - x) rounding.
    In place on P, Q, Bi; returns the sum of 0.5*error**2."""
    assert P.dtype == np.float64 and Q.dtype == np.float32 and Bi.dtype == np.float64
    m = len(user_ptr) - 1
    nus = np.diff(user_ptr)
    neg_ptr = np.concatenate([[0], np.cumsum(np.where(nus > 1, nus * rho, 0))])
    half_sq = 0
    for u0 in range(0, m, round_users):
        u1 = min(m, u0 + round_users)
        dP, dQ, dB = np.zeros_like(P), np.zeros_like(Q), np.zeros_like(Bi)
        for u in range(u0, u1):
            Pw, Qw, Bw = P.copy(), Q.copy(), Bi.copy()           # the user's private view (small problems only: this is the checker)
            half_sq += fism_epoch(Pw, Qw, Bw, np.array([0, nus[u]]), ev_i[user_ptr[u]:user_ptr[u + 1]], negs[neg_ptr[u]:neg_ptr[u + 1]],
                                  rho, alpha, lRate, regI, regB)
            dP += Pw - P
            dQ += Qw - Q
            dB += Bw - Bi
        P += dP
        Q += dQ
        Bi += dB
    return half_sq
