"""CPU, world_size 2: the control plane bench.py uses (yue_amd/dist.py: standard-library TCP star -- id broadcast,
barrier, max) and the sharded S-round epoch (executable spec with oracle arithmetic): the all-reduced user
factors are identical on both ranks and equal a single-process emulation of the two shards.  Plus the host-side
schedule of yue_bpr_epoch (user blocks, all-reduce groups) at BASELINE config 4's counts for 8 ranks, taken from
the library itself (yue_epoch_plan: pure host arithmetic, no GPU) against yue_amd/dist.py."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from helpers.sharded_spec import epoch_spec, shard_problem

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_two_ranks(tmp_path, through_torchrun):
    script = os.path.join(ROOT, 'tests', 'helpers', 'rank_main.py')
    port = _free_port()
    if through_torchrun:
        # as the driver launches bench.py: the launcher's own store sits on MASTER_PORT, the ranks must not collide with it
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
               '--master-addr', '127.0.0.1', '--master-port', str(port), script, str(tmp_path)]
        res = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, OMP_NUM_THREADS='1'), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        assert res.returncode == 0, res.stdout.decode()[-3000:]
        return
    procs = []
    for r in range(2):
        env = dict(os.environ, OMP_NUM_THREADS='1', RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, script, str(tmp_path)], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=600)
        assert p.returncode == 0, out.decode()[-3000:]


@pytest.mark.parametrize('through_torchrun', [False, True])
def test_two_rank_sharded_epoch_over_the_tcp_control_plane(tmp_path, orc, through_torchrun):
    _run_two_ranks(tmp_path, through_torchrun)
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


def epoch_spec_blocks(orc, store, rank, data, P, Q, seed, epoch, round_events, lr, regU, regI, events_total, world=2):
    """epoch_spec with the all-reduce replaced by recording this rank's block differences; valid
    because a block's users are not read again within the epoch (users never straddle blocks)."""
    from yue_amd.dist import user_block_width
    ub = user_block_width(round_events, events_total, P.shape[0], world)
    pos = [0]

    def record(block):
        u0 = pos[0]
        store.append((u0, u0 + len(block), block.copy()))
        pos[0] += ub
        block[:] = 0                      # leave P untouched in this pass
    epoch_spec(orc, record, world, rank, data, P, Q, seed, epoch, round_events, lr, regU, regI, events_total)


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


def test_epoch_schedule_of_the_library_for_eight_ranks_on_config4():
    # BASELINE config 4: 10M users, 500M events over 8 item shards, k = 128; the library's host arithmetic
    # (yue_hip.hip: yue_epoch_plan, used by yue_bpr_epoch on every rank) against the Python restatement
    from yue_amd._shim import epoch_plan
    from yue_amd.dist import epoch_block_plan, user_block_width
    m, k, etot = 10000000, 128, 500e6
    for world in (1, 2, 4, 8):
        for W in (49152, 32768, 1000, 7):
            ub, grp, nb = epoch_plan(m, k, W, etot, world)
            plan = epoch_block_plan(m, k, W, etot, world)
            assert ub == user_block_width(W, etot, m, world) == plan['user_block']
            assert grp == plan['blocks_per_group'] and nb == plan['n_blocks']
            groups = plan['groups']
            assert groups[0][0] == 0 and groups[-1][1] == m
            assert all(a[1] == b[0] for a, b in zip(groups, groups[1:]))           # contiguous, no user twice
            assert all(g[2] == (g[1] - g[0]) * k for g in groups)
            if len(groups) > 1:
                assert min(g[2] * 4 for g in groups[:-1]) >= (8 << 20) - ub * k * 4    # every full group carries about 8 MB or more
    # 8 ranks, default round: 6.25 events per user and rank -> 7,864 users per round, 2 rounds per all-reduce
    ub, grp, nb = epoch_plan(m, k, 49152, etot, 8)
    assert (ub, grp, nb) == (7864, 2, 1272)
    # a user is never split: block boundaries are user boundaries for any ragged event list
    from yue_amd.dist import epoch_round_ptr
    rs = np.random.RandomState(3)
    ev_ptr = np.concatenate([[0], np.cumsum(rs.randint(0, 9, size=1000))])
    rp = epoch_round_ptr(ev_ptr, 64, events_total=float(ev_ptr[-1]) * 8, world=8)
    assert rp[0] == 0 and rp[-1] == ev_ptr[-1] and set(rp) <= set(ev_ptr.tolist())
