"""Executable specification (CPU, oracle arithmetic) of what yue_bpr_epoch does on a communicator:
items sharded over ranks, users replicated, rounds = identical user blocks on every rank, the
block's user-factor differences summed over ranks (all-reduce) and applied everywhere, item-factor
differences applied locally.  Test infrastructure: used by tests/test_dist_cpu.py (2 gloo ranks)
and tests/test_gpu_multi.py (the GPU path with a 1-rank communicator)."""
import numpy as np

from yue_amd import synth
from yue_amd.dist import user_block_width

SEED_RANK = 0x632BE59BD9B4E019       # per-shard sampler stream, as in csrc/yue_hip.hip


def shard_problem(rank, m, n_local, d, k):
    data = synth.make_arrays(m, n_local, d, seed=100 + rank)
    P0, _ = synth.init_factors(m, n_local, k, 9)
    Q0 = synth.init_factors(1, n_local, k, 9 + 17 * (rank + 1))[1]
    return data, P0, Q0


def rank_seed(seed, rank):
    return (seed + SEED_RANK * rank) & 0xFFFFFFFFFFFFFFFF


def epoch_spec(orc, allreduce, world, rank, data, P, Q, seed, epoch, round_events, lr, regU, regI, events_total):
    """One epoch on this rank's shard, in place on P (replicated) and Q (local).  `allreduce(dP_block)`
    sums a float32 array over ranks in place.  Returns the local nll."""
    m, n = P.shape[0], Q.shape[0]
    ev_ptr = data['ev_ptr']
    ev_u = np.repeat(np.arange(m, dtype=np.int32), np.diff(ev_ptr))
    j = orc.sample_counter(rank_seed(seed, rank), epoch, ev_u, n, data['indptr'], data['indices'])
    ub = user_block_width(round_events, events_total, m, world)
    nll = 0.0
    for u0 in range(0, m, ub):
        u1 = min(m, u0 + ub)
        e0, e1 = int(ev_ptr[u0]), int(ev_ptr[u1])
        part, dP, dQ = orc.bpr_round_deltas(P, Q, ev_u[e0:e1], data['ev_i'][e0:e1], j[e0:e1], lr, regU, regI)
        nll += part
        Q += dQ
        block = np.ascontiguousarray(dP[u0:u1])
        allreduce(block)
        P[u0:u1] += block
    return nll
