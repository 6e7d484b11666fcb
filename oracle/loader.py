"""ctypes front end of oracle/liboracle.so (test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))


def lib_path():
    return os.path.join(_DIR, 'liboracle.so')


def build(force=False):
    src = os.path.join(_DIR, 'bpr_oracle.c')
    so = lib_path()
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-s', '-C', _DIR, '-B', 'liboracle.so'])
    return so


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class Oracle(object):
    def __init__(self):
        self.lib = C.CDLL(build())
        L = self.lib
        L.orc_sample_python.restype = C.c_int64
        L.orc_bpr_sequential.restype = C.c_double
        L.orc_bpr_rounds.restype = C.c_double
        L.orc_bpr_hogwild.restype = C.c_double
        L.orc_sumsq.restype = C.c_double
        L.orc_bpr_round_deltas.restype = C.c_double
        L.orc_topn_scan.restype = C.c_int
        L.orc_dependency_depth.restype = C.c_int64
        L.orc_bpr_rounds_seq_user.restype = C.c_double
        L.orc_dataflow_model.restype = C.c_double

    # -- samplers -------------------------------------------------------------
    def sample_python(self, seed, epochs, ev_u, n, indptr, indices, want_draws=False):
        ev_u = np.ascontiguousarray(ev_u, np.int32)
        indptr = np.ascontiguousarray(indptr, np.int64)
        indices = np.ascontiguousarray(indices, np.int32)
        E = len(ev_u)
        j = np.empty(epochs * E, np.int32)
        cap = 4 * epochs * E + 1024 if want_draws else 0
        draws = np.empty(max(cap, 1), np.int32)
        nd = self.lib.orc_sample_python(C.c_uint64(seed), C.c_int(epochs), _p(ev_u, C.c_int32), C.c_int64(E), C.c_int32(n),
                                        _p(indptr, C.c_int64), _p(indices, C.c_int32), _p(j, C.c_int32),
                                        _p(draws, C.c_int32) if want_draws else None, C.c_int64(cap))
        return (j, draws[:min(nd, cap)]) if want_draws else j

    def sample_counter(self, seed, epoch, ev_u, n_range, indptr, indices, e0=0, lo=0):
        ev_u = np.ascontiguousarray(ev_u, np.int32)
        indptr = np.ascontiguousarray(indptr, np.int64)
        indices = np.ascontiguousarray(indices, np.int32)
        j = np.empty(len(ev_u), np.int32)
        self.lib.orc_sample_counter(C.c_uint64(seed), C.c_uint32(epoch), _p(ev_u, C.c_int32), C.c_int64(e0), C.c_int64(len(ev_u)),
                                    C.c_int32(lo), C.c_int32(n_range), _p(indptr, C.c_int64), _p(indices, C.c_int32), _p(j, C.c_int32))
        return j

    # -- training ---------------------------------------------------------------
    def bpr_sequential(self, P, Q, u, i, j, lr, regU, regI):
        """In place on P, Q (float32 C-contiguous). Returns sum(-log s)."""
        assert P.dtype == np.float32 and Q.dtype == np.float32 and P.flags.c_contiguous and Q.flags.c_contiguous
        u, i, j = (np.ascontiguousarray(x, np.int32) for x in (u, i, j))
        return self.lib.orc_bpr_sequential(_p(P, C.c_float), _p(Q, C.c_float), C.c_int(P.shape[1]), _p(u, C.c_int32), _p(i, C.c_int32),
                                           _p(j, C.c_int32), C.c_int64(len(u)), C.c_double(lr), C.c_double(regU), C.c_double(regI))

    def bpr_hogwild(self, P, Q, u, i, j, lr, regU, regI, threads):
        """Timing baseline: the sequential loop raced by `threads` threads over slices of the stream."""
        assert P.dtype == np.float32 and Q.dtype == np.float32 and P.flags.c_contiguous and Q.flags.c_contiguous
        u, i, j = (np.ascontiguousarray(x, np.int32) for x in (u, i, j))
        return self.lib.orc_bpr_hogwild(_p(P, C.c_float), _p(Q, C.c_float), C.c_int(P.shape[1]), _p(u, C.c_int32), _p(i, C.c_int32),
                                        _p(j, C.c_int32), C.c_int64(len(u)), C.c_double(lr), C.c_double(regU), C.c_double(regI), C.c_int(threads))

    def dependency_depth(self, u, i, j, m, n):
        """(longest chain of dependent triplets, touches of the hottest item row) of a stream under sequential semantics."""
        u, i, j = (np.ascontiguousarray(x, np.int32) for x in (u, i, j))
        row_max = C.c_int64()
        depth = self.lib.orc_dependency_depth(_p(u, C.c_int32), _p(i, C.c_int32), _p(j, C.c_int32), C.c_int64(len(u)), C.c_int64(m), C.c_int64(n), C.byref(row_max))
        return int(depth), int(row_max.value)

    def dataflow_model(self, u, i, j, n, workers, startup, step, hop, hop_same):
        """Timing model of the exact path's dataflow launch (see orc_dataflow_model): (makespan, cross-run hops on its longest path)."""
        u, i, j = (np.ascontiguousarray(x, np.int32) for x in (u, i, j))
        hops = C.c_int64()
        t = self.lib.orc_dataflow_model(_p(u, C.c_int32), _p(i, C.c_int32), _p(j, C.c_int32), C.c_int64(len(u)), C.c_int64(n), C.c_int(workers),
                                        C.c_double(startup), C.c_double(step), C.c_double(hop), C.c_double(hop_same), C.byref(hops))
        return float(t), int(hops.value)

    def bpr_rounds(self, P, Q, u, i, j, round_ptr, lr, regU, regI):
        assert P.dtype == np.float32 and Q.dtype == np.float32 and P.flags.c_contiguous and Q.flags.c_contiguous
        u, i, j = (np.ascontiguousarray(x, np.int32) for x in (u, i, j))
        rp = np.ascontiguousarray(round_ptr, np.int64)
        return self.lib.orc_bpr_rounds(_p(P, C.c_float), _p(Q, C.c_float), C.c_int64(P.shape[0]), C.c_int64(Q.shape[0]), C.c_int(P.shape[1]),
                                       _p(u, C.c_int32), _p(i, C.c_int32), _p(j, C.c_int32), _p(rp, C.c_int64), C.c_int64(len(rp) - 1),
                                       C.c_double(lr), C.c_double(regU), C.c_double(regI))

    def bpr_rounds_seq_user(self, P, Q, u, i, j, round_ptr, lr, regU, regI):
        assert P.dtype == np.float32 and Q.dtype == np.float32 and P.flags.c_contiguous and Q.flags.c_contiguous
        u, i, j = (np.ascontiguousarray(x, np.int32) for x in (u, i, j))
        rp = np.ascontiguousarray(round_ptr, np.int64)
        return self.lib.orc_bpr_rounds_seq_user(_p(P, C.c_float), _p(Q, C.c_float), C.c_int64(P.shape[0]), C.c_int64(Q.shape[0]), C.c_int(P.shape[1]),
                                       _p(u, C.c_int32), _p(i, C.c_int32), _p(j, C.c_int32), _p(rp, C.c_int64), C.c_int64(len(rp) - 1),
                                       C.c_double(lr), C.c_double(regU), C.c_double(regI))

    def bpr_round_deltas(self, P, Q, u, i, j, lr, regU, regI):
        """(nll, dP, dQ): summed per-row differences of one round, factors untouched."""
        u, i, j = (np.ascontiguousarray(x, np.int32) for x in (u, i, j))
        dP = np.zeros_like(P)
        dQ = np.zeros_like(Q)
        nll = self.lib.orc_bpr_round_deltas(_p(P, C.c_float), _p(Q, C.c_float), C.c_int(P.shape[1]), _p(u, C.c_int32), _p(i, C.c_int32),
                                            _p(j, C.c_int32), C.c_int64(len(u)), C.c_double(lr), C.c_double(regU), C.c_double(regI),
                                            _p(dP, C.c_float), _p(dQ, C.c_float))
        return nll, dP, dQ

    def sumsq(self, X):
        X = np.ascontiguousarray(X, np.float32)
        return self.lib.orc_sumsq(_p(X, C.c_float), C.c_int64(X.size))

    # -- scoring ------------------------------------------------------------------
    def scores(self, P, Q, user):
        out = np.empty(Q.shape[0], np.float32)
        self.lib.orc_scores(_p(P, C.c_float), _p(Q, C.c_float), C.c_int64(Q.shape[0]), C.c_int(Q.shape[1]), C.c_int32(user), _p(out, C.c_float))
        return out

    def _topn(self, fn, P, Q, users, N, mask_indptr, mask_indices):
        users = np.ascontiguousarray(users, np.int32)
        mp = np.ascontiguousarray(mask_indptr, np.int64)
        mi = np.ascontiguousarray(mask_indices, np.int32)
        if mi.size == 0:
            mi = np.zeros(1, np.int32)
        ids = np.empty((len(users), N), np.int32)
        sc = np.empty((len(users), N), np.float32)
        rc = fn(_p(P, C.c_float), _p(Q, C.c_float), C.c_int64(Q.shape[0]), C.c_int(Q.shape[1]), _p(users, C.c_int32), C.c_int64(len(users)),
                C.c_int(N), _p(mp, C.c_int64), _p(mi, C.c_int32), _p(ids, C.c_int32), _p(sc, C.c_float))
        return ids, sc, rc

    def topn_scan(self, P, Q, users, N, mask_indptr, mask_indices):
        return self._topn(self.lib.orc_topn_scan, P, Q, users, N, mask_indptr, mask_indices)

    def topn_true(self, P, Q, users, N, mask_indptr, mask_indices):
        return self._topn(self.lib.orc_topn_true, P, Q, users, N, mask_indptr, mask_indices)
