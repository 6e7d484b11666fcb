"""ctypes binding of libyue_hip.so (include/yue_hip.h).

This is the only route from the Python plugin surface to the numeric hot path: there is no
CPU fallback.  If the library is missing or no MI355X is visible, calls fail loudly in the
reference's style -- print the message, exit(-1) (tool/config.py:9-11,
base/IterativeRecommender.py:64-66) -- or raise YueHipError when ``raise_errors`` is set
(tests use that).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libyue_hip.so')

OK, ERR_ARG, ERR_HIP, ERR_FEW_ITEMS, ERR_COMM = 0, -1, -2, -3, -4
UNIQUE_ID_BYTES = 128

# every symbol include/yue_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = ['yue_last_error', 'yue_version', 'yue_ctx_create', 'yue_ctx_destroy', 'yue_sync',
           'yue_set_factors', 'yue_get_factors', 'yue_set_interactions', 'yue_bpr_replay',
           'yue_bpr_rounds', 'yue_bpr_epoch', 'yue_cune_steps', 'yue_adam_reset', 'yue_adam_step', 'yue_sample_negatives', 'yue_sumsq', 'yue_scores',
           'yue_topn_scan', 'yue_set_kernel_timing', 'yue_get_kernel_timing', 'yue_get_scan_stats', 'yue_get_scan_work', 'yue_set_option', 'yue_get_option',
           'yue_comm_unique_id', 'yue_comm_init', 'yue_allreduce_f64', 'yue_get_comm_stats',
           'yue_default_round_events', 'yue_epoch_plan',
           'yue_fism_set_model', 'yue_fism_get_model', 'yue_fism_epoch', 'yue_fism_rounds', 'yue_fism_scores', 'yue_fism_topn_scan']


class YueHipError(RuntimeError):
    def __init__(self, code, msg):
        super(YueHipError, self).__init__('libyue_hip error %d: %s' % (code, msg))
        self.code = code


_lib = None


def load_library():
    """dlopen libyue_hip.so; never falls back to anything else."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise YueHipError(ERR_HIP, 'HIP library not built: %s (run `make -C yue_amd/csrc` or __graft_entry__.build())' % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        lib.yue_last_error.restype = C.c_char_p
        for name in SYMBOLS[1:]:
            getattr(lib, name).restype = C.c_int
        _lib = lib
    return _lib


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def _i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(C.POINTER(C.c_int32))


def _i64(a):
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a, a.ctypes.data_as(C.POINTER(C.c_int64))


def _f64(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


def comm_unique_id():
    lib = load_library()
    buf = (C.c_ubyte * UNIQUE_ID_BYTES)()
    rc = lib.yue_comm_unique_id(buf)
    if rc != OK:
        raise YueHipError(rc, lib.yue_last_error().decode())
    return bytes(buf)


def epoch_plan(m, k, round_events, events_total, nranks):
    """(user_block, blocks_per_group, n_blocks) of yue_bpr_epoch's schedule -- pure host arithmetic inside the library."""
    lib = load_library()
    ub, grp, nb = C.c_int64(), C.c_int64(), C.c_int64()
    rc = lib.yue_epoch_plan(C.c_int64(m), C.c_int(k), C.c_int64(round_events), C.c_double(events_total), C.c_int(nranks),
                            C.byref(ub), C.byref(grp), C.byref(nb))
    if rc != OK:
        raise YueHipError(rc, lib.yue_last_error().decode())
    return ub.value, grp.value, nb.value


class Device(object):
    """One HIP context = one GPU (one process per GPU)."""

    def __init__(self, device=0, raise_errors=False):
        self._lib = load_library()
        self._raise = raise_errors
        self._ctx = C.c_void_p()
        self.m = self.n = self.k = self.E = 0
        self._chk(self._lib.yue_ctx_create(C.c_int(device), C.byref(self._ctx)))

    # reference convention: print, exit(-1)
    def _chk(self, rc):
        if rc == OK:
            return
        msg = self._lib.yue_last_error().decode()
        if self._raise:
            raise YueHipError(rc, msg)
        print(msg)
        exit(-1)

    def close(self):
        if self._ctx:
            self._lib.yue_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        self._chk(self._lib.yue_sync(self._ctx))

    # -- state ------------------------------------------------------------------------
    def set_factors(self, P, Q):
        P, pp = _f32(P)
        Q, qp = _f32(Q)
        assert P.ndim == 2 and Q.ndim == 2 and P.shape[1] == Q.shape[1]
        self.m, self.k = P.shape
        self.n = Q.shape[0]
        self._chk(self._lib.yue_set_factors(self._ctx, pp, C.c_int64(self.m), qp, C.c_int64(self.n), C.c_int(self.k)))

    def get_factors(self, P=None, Q=None):
        """Copies the device factors into P, Q (allocated when None) and returns them."""
        if P is None:
            P = np.empty((self.m, self.k), np.float32)
        if Q is None:
            Q = np.empty((self.n, self.k), np.float32)
        assert P.dtype == np.float32 and Q.dtype == np.float32 and P.flags.c_contiguous and Q.flags.c_contiguous
        self._chk(self._lib.yue_get_factors(self._ctx, P.ctypes.data_as(C.POINTER(C.c_float)), Q.ctypes.data_as(C.POINTER(C.c_float))))
        return P, Q

    def set_interactions(self, indptr, indices, ev_ptr, ev_i):
        indptr, a = _i64(indptr)
        indices, b = _i32(indices if len(indices) else np.zeros(1, np.int32))
        ev_ptr, c = _i64(ev_ptr)
        ev_i, d = _i32(ev_i if len(ev_i) else np.zeros(1, np.int32))
        assert len(indptr) == self.m + 1 and len(ev_ptr) == self.m + 1
        self.E = int(ev_ptr[-1])
        self._chk(self._lib.yue_set_interactions(self._ctx, a, b, c, d))

    # -- training -----------------------------------------------------------------------
    def bpr_replay(self, u, i, j, lr, regU, regI):
        u, a = _i32(u)
        i, b = _i32(i)
        j, c = _i32(j)
        nll = C.c_double()
        self._chk(self._lib.yue_bpr_replay(self._ctx, a, b, c, C.c_int64(len(u)), C.c_double(lr), C.c_double(regU), C.c_double(regI), C.byref(nll)))
        return nll.value

    def bpr_rounds(self, u, i, j, round_ptr, lr, regU, regI):
        u, a = _i32(u)
        i, b = _i32(i)
        j, c = _i32(j)
        rp, d = _i64(round_ptr)
        nll = C.c_double()
        self._chk(self._lib.yue_bpr_rounds(self._ctx, a, b, c, d, C.c_int64(len(rp) - 1), C.c_double(lr), C.c_double(regU), C.c_double(regI), C.byref(nll)))
        return nll.value

    def cune_steps(self, u, i, k, j, s, lr, regU, regI):
        """CUNE's two-level BPR steps in order (k < 0: plain step).  Returns the per-step losses (float64[T])."""
        u, a = _i32(u)
        i, b = _i32(i)
        k, c = _i32(k)
        j, d = _i32(j)
        loss = np.empty(len(u), np.float64)
        self._chk(self._lib.yue_cune_steps(self._ctx, a, b, c, d, C.c_int64(len(u)), C.c_double(s), C.c_double(lr), C.c_double(regU), C.c_double(regI),
                                           loss.ctypes.data_as(C.POINTER(C.c_double))))
        return loss

    def adam_reset(self):
        self._chk(self._lib.yue_adam_reset(self._ctx))

    def adam_step(self, u, i, j, lr, reg, step):
        """One minibatch step of the reference's live TF-style path (softplus loss + l2 terms, Adam).  Returns total_loss."""
        u, a = _i32(u)
        i, b = _i32(i)
        j, c = _i32(j)
        loss = C.c_double()
        self._chk(self._lib.yue_adam_step(self._ctx, a, b, c, C.c_int64(len(u)), C.c_double(lr), C.c_double(reg), C.c_int64(step), C.byref(loss)))
        return loss.value

    def default_round_events(self):
        out = C.c_int64()
        self._chk(self._lib.yue_default_round_events(self._ctx, C.byref(out)))
        return out.value

    def bpr_epoch(self, seed, epoch, round_events, lr, regU, regI):
        """Returns (nll, sumsqP, sumsqQ) after one epoch (device sampler + S-rounds; round_events 0 = the device default)."""
        nll, sp, sq = C.c_double(), C.c_double(), C.c_double()
        self._chk(self._lib.yue_bpr_epoch(self._ctx, C.c_uint64(seed), C.c_uint32(epoch), C.c_int64(round_events), C.c_double(lr), C.c_double(regU),
                                          C.c_double(regI), C.byref(nll), C.byref(sp), C.byref(sq)))
        return nll.value, sp.value, sq.value

    def sample_negatives(self, seed, epoch):
        j = np.empty(max(self.E, 1), np.int32)
        self._chk(self._lib.yue_sample_negatives(self._ctx, C.c_uint64(seed), C.c_uint32(epoch), j.ctypes.data_as(C.POINTER(C.c_int32))))
        return j[:self.E]

    def sumsq(self):
        sp, sq = C.c_double(), C.c_double()
        self._chk(self._lib.yue_sumsq(self._ctx, C.byref(sp), C.byref(sq)))
        return sp.value, sq.value

    # -- scoring ------------------------------------------------------------------------
    def scores(self, user):
        out = np.empty(self.n, np.float32)
        self._chk(self._lib.yue_scores(self._ctx, C.c_int32(user), out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def topn_scan(self, users, N, mask_indptr=None, mask_indices=None):
        """(ids[nu,N] int32, scores[nu,N] float32).  Raises IndexError like the reference
        (base/IterativeRecommender.py:126) when a user has fewer than N candidates."""
        users, up = _i32(users)
        ids = np.empty((len(users), N), np.int32)
        sc = np.empty((len(users), N), np.float32)
        if mask_indptr is None:
            mp = mi = None
        else:
            mask_indptr, mp = _i64(mask_indptr)
            mask_indices, mi = _i32(mask_indices if len(mask_indices) else np.zeros(1, np.int32))
        rc = self._lib.yue_topn_scan(self._ctx, up, C.c_int64(len(users)), C.c_int(N), mp, mi,
                                     ids.ctypes.data_as(C.POINTER(C.c_int32)), sc.ctypes.data_as(C.POINTER(C.c_float)))
        if rc == ERR_FEW_ITEMS:
            raise IndexError('list index out of range')
        self._chk(rc)
        return ids, sc

    def scan_stats(self):
        """(kernel ms, state-machine events, exact re-scores, used_bf16) of the last topn_scan."""
        ms, ev, rs, bf = C.c_double(), C.c_int64(), C.c_int64(), C.c_int()
        self._chk(self._lib.yue_get_scan_stats(self._ctx, C.byref(ms), C.byref(ev), C.byref(rs), C.byref(bf)))
        return ms.value, ev.value, rs.value, bool(bf.value)

    def scan_work(self):
        """(tiles scored, tiles of the full product) of the last topn_scan (32 users x 32 items each)."""
        done, total = C.c_int64(), C.c_int64()
        self._chk(self._lib.yue_get_scan_work(self._ctx, C.byref(done), C.byref(total)))
        return done.value, total.value

    def set_option(self, name, value):
        self._chk(self._lib.yue_set_option(self._ctx, name.encode(), C.c_int64(value)))

    def get_option(self, name):
        out = C.c_int64()
        self._chk(self._lib.yue_get_option(self._ctx, name.encode(), C.byref(out)))
        return out.value

    # -- measurement ----------------------------------------------------------------------
    def set_kernel_timing(self, stride):
        self._chk(self._lib.yue_set_kernel_timing(self._ctx, C.c_int(stride)))

    def get_kernel_timing(self):
        ms, nl, nt = C.c_double(), C.c_int64(), C.c_int64()
        self._chk(self._lib.yue_get_kernel_timing(self._ctx, C.byref(ms), C.byref(nl), C.byref(nt)))
        return ms.value, nl.value, nt.value

    # -- multi-GPU ----------------------------------------------------------------------
    # -- FISM (parity path) ---------------------------------------------------------------
    def fism_set_model(self, P, Q, Bi):
        P, a = _f64(P)
        Q, b = _f32(Q)
        Bi, c = _f64(Bi)
        assert P.shape == Q.shape and len(Bi) == P.shape[0]
        self.fn, self.fk = P.shape
        self._chk(self._lib.yue_fism_set_model(self._ctx, a, b, c, C.c_int64(self.fn), C.c_int(self.fk)))

    def fism_get_model(self, P, Q, Bi):
        """Copies the device model into the given arrays (float64 [n,k], float32 [n,k], float64 [n])."""
        assert P.dtype == np.float64 and Q.dtype == np.float32 and Bi.dtype == np.float64
        assert P.flags.c_contiguous and Q.flags.c_contiguous and Bi.flags.c_contiguous
        self._chk(self._lib.yue_fism_get_model(self._ctx, P.ctypes.data_as(C.POINTER(C.c_double)), Q.ctypes.data_as(C.POINTER(C.c_float)),
                                               Bi.ctypes.data_as(C.POINTER(C.c_double))))

    def fism_epoch(self, user_ptr, ev_i, negs, rho, coef, lr, regI, regB):
        """One sequential pass (FISM.py:38-69).  Returns (sum of 0.5*error^2, sum(P*P), sum(Q*Q), Bi.Bi)."""
        user_ptr, a = _i64(user_ptr)
        ev_i, b = _i32(ev_i if len(ev_i) else np.zeros(1, np.int32))
        n_negs = len(negs)
        negs, c = _i32(negs if n_negs else np.zeros(1, np.int32))
        coef, d = _f64(coef)
        half = C.c_double()
        sums = (C.c_double * 3)()
        self._chk(self._lib.yue_fism_epoch(self._ctx, a, C.c_int64(len(user_ptr) - 1), b, c, C.c_int64(n_negs), C.c_int(rho), d,
                                           C.c_double(lr), C.c_double(regI), C.c_double(regB), C.byref(half), sums))
        return half.value, sums[0], sums[1], sums[2]

    def fism_rounds(self, user_ptr, ev_i, negs, rho, coef, round_users, lr, regI, regB):
        """The same pass in rounds of `round_users` users (throughput form).  Returns (sum of 0.5*error^2, sum(P*P), sum(Q*Q), Bi.Bi)."""
        user_ptr, a = _i64(user_ptr)
        ev_i, b = _i32(ev_i if len(ev_i) else np.zeros(1, np.int32))
        n_negs = len(negs)
        negs, c = _i32(negs if n_negs else np.zeros(1, np.int32))
        coef, d = _f64(coef)
        half = C.c_double()
        sums = (C.c_double * 3)()
        self._chk(self._lib.yue_fism_rounds(self._ctx, a, C.c_int64(len(user_ptr) - 1), b, c, C.c_int64(n_negs), C.c_int(rho), d, C.c_int64(round_users),
                                            C.c_double(lr), C.c_double(regI), C.c_double(regB), C.byref(half), sums))
        return half.value, sums[0], sums[1], sums[2]

    def fism_scores(self, items):
        items, a = _i32(items if len(items) else np.zeros(1, np.int32))
        out = np.empty(self.fn, np.float64)
        self._chk(self._lib.yue_fism_scores(self._ctx, a, C.c_int64(len(items)), out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def fism_topn_scan(self, row_ptr, row_items, N):
        """(ids[nu,N] int32, scores[nu,N] float64) for users given by the CSR of their training events."""
        row_ptr, a = _i64(row_ptr)
        nu = len(row_ptr) - 1
        row_items, b = _i32(row_items if len(row_items) else np.zeros(1, np.int32))
        ids = np.empty((nu, N), np.int32)
        sc = np.empty((nu, N), np.float64)
        rc = self._lib.yue_fism_topn_scan(self._ctx, a, b, C.c_int64(nu), C.c_int(N), ids.ctypes.data_as(C.POINTER(C.c_int32)),
                                          sc.ctypes.data_as(C.POINTER(C.c_double)))
        if rc == ERR_FEW_ITEMS:
            raise IndexError(self._lib.yue_last_error().decode())
        self._chk(rc)
        return ids, sc

    def comm_init(self, unique_id, rank, nranks):
        buf = (C.c_ubyte * UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
        self._chk(self._lib.yue_comm_init(self._ctx, buf, C.c_int(rank), C.c_int(nranks)))

    def comm_stats(self):
        """Last bpr_epoch on a communicator: dict(allreduce_bytes, collectives, wait_ms, nranks, rccl_version)."""
        b, n, w, r, v = C.c_double(), C.c_int64(), C.c_double(), C.c_int(), C.c_int()
        self._chk(self._lib.yue_get_comm_stats(self._ctx, C.byref(b), C.byref(n), C.byref(w), C.byref(r), C.byref(v)))
        return {'allreduce_bytes': b.value, 'collectives': n.value, 'wait_ms': w.value, 'nranks': r.value, 'rccl_version': v.value}

    def allreduce_f64(self, vals):
        arr = (C.c_double * len(vals))(*vals)
        self._chk(self._lib.yue_allreduce_f64(self._ctx, arr, C.c_int(len(vals))))
        return list(arr)
