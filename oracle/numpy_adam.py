"""NumPy restatement of the reference's LIVE BPR path -- the TensorFlow-1 graph of recommender/cf/BPR.py:83-129
(embedding lookups, softplus loss, l2 terms, AdamOptimizer) -- TEST INFRASTRUCTURE.  SURVEY 8(a) row a9.

PARITY UNPINNED: TensorFlow is not installable here, so nothing in this file has been checked against the reference's
own execution; it restates the graph as written in the reference and the documented behaviour of the TF-1 ops it names:
  * tf.nn.embedding_lookup gradients are IndexedSlices; the optimizer sums duplicate indices first;
  * tf.nn.l2_loss(x) = sum(x**2) / 2 over every gathered row (a row gathered 100 times counts 100 times);
  * tf.train.AdamOptimizer (beta1 0.9, beta2 0.999, epsilon 1e-8) on IndexedSlices: m <- beta1*m, then the summed
    gradient rows are scatter-added with (1-beta1); v likewise with the squares; var <- var - lr_t * m / (sqrt(v) + eps)
    for ALL rows, lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t) -- i.e. dense Adam with a zero gradient on untouched rows;
  * everything in float32.
The device path (yue_adam_step) is compared with THIS restatement (tests/test_gpu_adam.py).
"""
import numpy as np

BETA1, BETA2, EPS = np.float32(0.9), np.float32(0.999), np.float32(1e-8)


def softplus(x):
    x = x.astype(np.float32)
    return (np.maximum(x, 0) + np.log1p(np.exp(-np.abs(x)))).astype(np.float32)


def sigmoid(x):
    x = x.astype(np.float32)
    return (np.float32(1) / (np.float32(1) + np.exp(-x))).astype(np.float32)


def adam_step(U, V, state, u_idx, i_idx, j_idx, lr, reg, t):
    """One `sess.run(train, total_loss)` of BPR.py:124 on the fed (u, i, j) triplets.  In place on U [m,k], V [n,k]
    (float32) and on state = dict(mU, vU, mV, vV), zeros before the first step; t = 1, 2, ... Returns total_loss."""
    reg = np.float32(reg)
    Ue, Ve, Ne = U[u_idx], V[i_idx], V[j_idx]
    err = (Ue * Ve).sum(axis=1) - (Ue * Ne).sum(axis=1)                      # :101
    loss = softplus(-err).sum(dtype=np.float32)                              # :102
    loss = loss + reg * ((Ue * Ue).sum(dtype=np.float32) + (Ve * Ve).sum(dtype=np.float32) + (Ne * Ne).sum(dtype=np.float32)) / np.float32(2)     # :104-107
    g = -sigmoid(-err)                                                       # d softplus(-e) / d e
    gU = np.zeros_like(U)
    gV = np.zeros_like(V)
    np.add.at(gU, u_idx, g[:, None] * (Ve - Ne) + reg * Ue)
    np.add.at(gV, i_idx, g[:, None] * Ue + reg * Ve)
    np.add.at(gV, j_idx, -g[:, None] * Ue + reg * Ne)
    lr_t = np.float32(lr * np.sqrt(1.0 - float(BETA2) ** t) / (1.0 - float(BETA1) ** t))
    for var, grad, m, v in ((U, gU, state['mU'], state['vU']), (V, gV, state['mV'], state['vV'])):
        m *= BETA1
        m += (np.float32(1) - BETA1) * grad
        v *= BETA2
        v += (np.float32(1) - BETA2) * grad * grad
        var -= lr_t * m / (np.sqrt(v) + EPS)
    return float(loss)


def new_state(U, V):
    return {'mU': np.zeros_like(U), 'vU': np.zeros_like(U), 'mV': np.zeros_like(V), 'vV': np.zeros_like(V)}


def truncated_normal(rs, shape, stddev):
    """tf.truncated_normal: normal(0, stddev), values beyond two standard deviations are drawn again (the stream of a
    NumPy RandomState -- TF's own generator is not reproducible here)."""
    x = rs.normal(0.0, stddev, size=shape)
    bad = np.abs(x) > 2 * stddev
    while bad.any():
        x[bad] = rs.normal(0.0, stddev, size=int(bad.sum()))
        bad = np.abs(x) > 2 * stddev
    return x.astype(np.float32)
