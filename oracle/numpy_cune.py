"""NumPy restatement of CUNE's two-level BPR training loop (reference recommender/advanced/CUNE.py:120-178) --
TEST INFRASTRUCTURE.  SURVEY 8(f) rank 3.

Only the SGD loop is restated: the collaborative user network, the random walks and the Word2Vec embedding that
produce the friends' item sets (CUNE.py:34-118) need gensim, which is absent here (parity unpinned for that stage);
the loop takes the friends' item sets as an input.  Every statement is the NumPy / math operation the reference
issues -- float32 rows, python-float coefficients under NEP-50 promotion, the sigmoid re-evaluated on the
already-updated rows in every statement, the loss that becomes a float32 once the per-user regulariser is added --
so the results equal the reference's bit for bit (tests/test_cune_golden.py pins them to tests/golden/g9_*).
"""
from math import exp, log

import numpy as np


def sigmoid(val):
    return 1 / (1 + exp(-val))                         # tool/qmath.py:115-116


def cune_step(P, Q, u, i, k, j, s, lRate, regU, regI):
    """One pass of the inner statements for a drawn (k, j); k < 0 is the branch for a user without friends' items
    (:163-169).  In place on P, Q (float32).  Returns this step's part of the loss (a python float)."""
    if k >= 0:
        P[u] += lRate * (1 - sigmoid(P[u].dot(Q[i]) - P[u].dot(Q[k]))) * (Q[i] - Q[k])                     # :133-134
        Q[i] += lRate * (1 - sigmoid(P[u].dot(Q[i]) - P[u].dot(Q[k]))) * P[u]                              # :135-136
        Q[k] -= lRate * (1 - sigmoid(P[u].dot(Q[i]) - P[u].dot(Q[k]))) * P[u]                              # :137-138
        P[u] += (1 / s) * lRate * (1 - sigmoid((1 / s) * (P[u].dot(Q[k]) - P[u].dot(Q[j])))) * (Q[k] - Q[j])     # :148-150
        Q[k] += (1 / s) * lRate * (1 - sigmoid((1 / s) * (P[u].dot(Q[k]) - P[u].dot(Q[j])))) * P[u]              # :151-152
        Q[j] -= (1 / s) * lRate * (1 - sigmoid((1 / s) * (P[u].dot(Q[k]) - P[u].dot(Q[j])))) * P[u]              # :153-154
        P[u] -= lRate * regU * P[u]                                                                        # :156-159
        Q[i] -= lRate * regI * Q[i]
        Q[j] -= lRate * regI * Q[j]
        Q[k] -= lRate * regI * Q[k]
        return -log(sigmoid(P[u].dot(Q[i]) - P[u].dot(Q[k]))) - log(sigmoid((1 / s) * (P[u].dot(Q[k]) - P[u].dot(Q[j]))))     # :161-162
    P[u] += lRate * (1 - sigmoid(P[u].dot(Q[i]) - P[u].dot(Q[j]))) * (Q[i] - Q[j])                         # :168
    Q[i] += lRate * (1 - sigmoid(P[u].dot(Q[i]) - P[u].dot(Q[j]))) * P[u]                                  # :169
    Q[j] -= lRate * (1 - sigmoid(P[u].dot(Q[i]) - P[u].dot(Q[j]))) * P[u]                                  # :170
    return -log(sigmoid(P[u].dot(Q[i]) - P[u].dot(Q[j])))                                                  # :172


def cune_epoch(P, Q, u, i, k, j, s, lRate, regU, regI):
    """One iteration of the while loop (:121-175) over an explicit (u, i, k, j) stream in the reference's order (three
    steps per event; the steps of a user are consecutive).  The regulariser is added once PER USER (:175 sits inside the
    user loop), which also turns the running loss into a NumPy float32 scalar from the first user on."""
    assert P.dtype == np.float32 and Q.dtype == np.float32
    loss = 0
    T = len(u)
    for t in range(T):
        loss += cune_step(P, Q, int(u[t]), int(i[t]), int(k[t]), int(j[t]), s, lRate, regU, regI)
        if t + 1 == T or u[t + 1] != u[t]:
            loss += regU * (P * P).sum() + regI * (Q * Q).sum()
    return loss
