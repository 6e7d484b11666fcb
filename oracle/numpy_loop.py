"""NumPy restatement of the reference's per-triplet loop -- TEST INFRASTRUCTURE (see oracle/__init__.py).

What recommender/cf/BPR.py:50-58 does for one (user, i, j), through the same NumPy operations
(``ndarray.dot`` -> BLAS sdot on fp32 rows, in-place fp32 row updates, python-float coefficients), so on
the same NumPy build it lands on the reference's factors bit for bit.  Two uses: a second pin of the C
oracle (tests/test_oracle_golden.py) and bench.py's "the loop as NumPy runs it" CPU rate.
"""
from math import exp, log

import numpy as np


def bpr_loop(P, Q, u, i, j, lRate, regU, regI):
    """In place on P, Q (float32 C-contiguous).  Returns the sum of -log(s) (BPR.py:58)."""
    assert P.dtype == np.float32 and Q.dtype == np.float32
    decay_u, decay_i = lRate * regU, lRate * regI       # python floats, as the reference forms them (:55-57)
    nll = 0
    for a, b, c in zip(u.tolist(), i.tolist(), j.tolist()):
        if c < 0:
            continue
        pu, qi, qj = P[a], Q[b], Q[c]                       # row views: the updates below write through
        margin = pu.dot(qi) - pu.dot(qj)                    # :50, fp32
        s = 1 / (1 + exp(-margin))                          # tool/qmath.py:115-116, double
        step = lRate * (1 - s)
        pu += step * (qi - qj)                              # :51
        qi += step * pu                                     # :52, with the updated user row
        qj -= step * pu                                     # :53
        pu -= decay_u * pu                                  # :55
        qi -= decay_i * qi                                  # :56
        qj -= decay_i * qj                                  # :57
        nll += -log(s)
    return nll
