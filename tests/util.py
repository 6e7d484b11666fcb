"""Shared helpers for the tests (host-side only)."""
import hashlib
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def gz(name):
    return np.load(os.path.join(GOLDEN, name))


def gj(name):
    return json.load(open(os.path.join(GOLDEN, name)))


def csr_from_events(ev_u, ev_i, m):
    """Sorted-unique item rows per user from an event list."""
    order = np.lexsort((ev_i, ev_u))
    u = np.asarray(ev_u)[order]
    i = np.asarray(ev_i)[order]
    keep = np.ones(len(u), bool)
    keep[1:] = (u[1:] != u[:-1]) | (i[1:] != i[:-1])
    u, i = u[keep], i[keep]
    indptr = np.zeros(m + 1, np.int64)
    np.add.at(indptr, u + 1, 1)
    return np.cumsum(indptr), i.astype(np.int32)


def ev_ptr_from_users(ev_u, m):
    """Event offsets per user; requires events grouped by ascending user id."""
    ev_u = np.asarray(ev_u)
    assert (np.diff(ev_u) >= 0).all()
    ptr = np.zeros(m + 1, np.int64)
    np.add.at(ptr, ev_u + 1, 1)
    return np.cumsum(ptr)


def mask_rows(indptr, indices, users):
    rows = [indices[indptr[x]:indptr[x + 1]] for x in users]
    mp = np.zeros(len(users) + 1, np.int64)
    mp[1:] = np.cumsum([len(r) for r in rows])
    mi = np.concatenate(rows).astype(np.int32) if rows else np.zeros(0, np.int32)
    return mp, mi


def rel_err(a, b):
    """max |a-b| / max |b| -- the '1e-5 rel fp32' yardstick of BASELINE.json."""
    return float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max() / np.abs(b).max())


def rel_err_elem(a, b, floor=1e-3):
    """Element-wise companion of rel_err: max |a-b| / |b| over the elements with |b| > floor
    (factors live in [0, 0.1): the floor keeps the near-zero elements, whose relative error is
    unbounded by construction, out of the quotient)."""
    a = a.astype(np.float64).ravel()
    b = b.astype(np.float64).ravel()
    big = np.abs(b) > floor
    if not big.any():
        return 0.0
    return float((np.abs(a[big] - b[big]) / np.abs(b[big])).max())
