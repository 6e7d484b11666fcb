"""CPU, world_size 2 over gloo: the control plane bench.py uses (id broadcast, barrier, max) and the
sharded S-round epoch (executable spec with oracle arithmetic): the all-reduced user factors are
identical on both ranks and equal a single-process emulation of the two shards."""
import os
import socket
import subprocess
import sys

import numpy as np

from helpers.sharded_spec import epoch_spec, shard_problem

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_rank_sharded_epoch_over_gloo(tmp_path, orc):
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
           os.path.join(ROOT, 'tests', 'helpers', 'rank_main.py'), str(tmp_path)]
    env = dict(os.environ, OMP_NUM_THREADS='1')
    res = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert res.returncode == 0, res.stdout.decode()[-3000:]
    r0 = np.load(tmp_path / 'rank0.npz')
    r1 = np.load(tmp_path / 'rank1.npz')
    assert np.array_equal(r0['P'], r1['P'])                     # replicated user factors stay identical
    assert not np.array_equal(r0['Q'], r1['Q'])

    # single-process emulation: same blocks, the two ranks' differences summed in rank order
    m, n_local, d, k = 300, 200, 12, 16
    shards = [shard_problem(r, m, n_local, d, k) for r in range(2)]
    P = shards[0][1].copy()
    Qs = [s[2].copy() for s in shards]
    etot = float(sum(s[0]['ev_ptr'][-1] for s in shards))
    assert etot == float(r0['events_total'])
    for epoch in range(2):
        # each rank's per-block differences against the same replicated P, then summed block by block
        blocks = [[], []]
        for r in range(2):
            epoch_spec_blocks(orc, blocks[r], r, shards[r][0], P.copy(), Qs[r], 77, epoch, 256, 0.05, 0.01, 0.01, etot)
        for (u0, u1, b0), (_, _, b1) in zip(blocks[0], blocks[1]):
            P[u0:u1] += b0 + b1
    assert np.array_equal(P, r0['P'])
    assert np.array_equal(Qs[0], r0['Q']) and np.array_equal(Qs[1], r1['Q'])


def epoch_spec_blocks(orc, store, rank, data, P, Q, seed, epoch, round_events, lr, regU, regI, events_total):
    """epoch_spec with the all-reduce replaced by recording this rank's block differences; valid
    because a block's users are not read again within the epoch (users never straddle blocks)."""
    from yue_amd.dist import user_block_width
    ub = user_block_width(round_events, events_total, P.shape[0], 2)
    pos = [0]

    def record(block):
        u0 = pos[0]
        store.append((u0, u0 + len(block), block.copy()))
        pos[0] += ub
        block[:] = 0                      # leave P untouched in this pass
    epoch_spec(orc, record, 2, rank, data, P, Q, seed, epoch, round_events, lr, regU, regI, events_total)


def test_one_rank_spec_equals_plain_rounds(orc):
    # world = 1: the sharded spec is the S-round oracle with rounds cut at user-block boundaries
    from yue_amd.dist import user_block_width
    m, n, d, k = 200, 150, 10, 8
    data, P, Q = shard_problem(0, m, n, d, k)
    P2, Q2 = P.copy(), Q.copy()
    E = float(data['ev_ptr'][-1])
    nll = epoch_spec(orc, lambda b: b, 1, 0, data, P, Q, 5, 0, 64, 0.05, 0.01, 0.01, E)
    ub = user_block_width(64, E, m, 1)
    ev_u = np.repeat(np.arange(m, dtype=np.int32), np.diff(data['ev_ptr']))
    j = orc.sample_counter(5, 0, ev_u, n, data['indptr'], data['indices'])
    rp = data['ev_ptr'][np.unique(np.concatenate([np.arange(0, m, ub), [m]]))]
    nll2 = orc.bpr_rounds(P2, Q2, ev_u, data['ev_i'], j, rp, 0.05, 0.01, 0.01)
    assert np.array_equal(P, P2) and np.array_equal(Q, Q2) and abs(nll - nll2) < 1e-9 * abs(nll2)
