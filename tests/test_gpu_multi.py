"""GPU: the communicator path of yue_bpr_epoch (user-aligned blocks, RCCL all-reduce of the block's
user-factor differences in place, range apply) with a 1-rank communicator on the one GPU we have,
against the executable spec that tests/test_dist_cpu.py runs on two CPU ranks; and the real thing --
two processes, two GPUs, RCCL -- wherever the box has two devices (skipped otherwise); and, on ANY box, two ranks on
one GPU with the collective behind the test seam of libyue_hip_seam.so (the block / group / two-stream logic of the
communicator path with more than one rank; RCCL itself still runs with one rank only on a one-GPU box)."""
import numpy as np
import pytest

from helpers.sharded_spec import epoch_spec, shard_problem
from util import rel_err

pytestmark = pytest.mark.gpu


def test_communicator_epoch_matches_sharded_spec(orc):
    from yue_amd._shim import Device, comm_unique_id
    m, n, d, k = 3000, 2000, 25, 128
    data, P0, Q0 = shard_problem(0, m, n, d, k)
    dev = Device(0, raise_errors=True)
    dev.set_factors(P0, Q0)
    dev.set_interactions(data['indptr'], data['indices'], data['ev_ptr'], data['ev_i'])
    dev.comm_init(comm_unique_id(), 0, 1)
    assert dev.allreduce_f64([1.5, 2.0]) == [1.5, 2.0]
    P, Q = P0.copy(), Q0.copy()
    E = float(data['ev_ptr'][-1])
    for epoch in range(2):
        nll, sp, sq = dev.bpr_epoch(31, epoch, 4096, 0.05, 0.01, 0.01)
        nll_s = epoch_spec(orc, lambda b: b, 1, 0, data, P, Q, 31, epoch, 4096, 0.05, 0.01, 0.01, E)
        Pg, Qg = dev.get_factors()
        assert rel_err(Pg, P) < 1e-5 and rel_err(Qg, Q) < 1e-5
        assert abs(nll - nll_s) <= 1e-9 * abs(nll_s)
    dev.close()


def _shard_reference(orc, m, n, d, k, W, epochs, nranks=2):
    """Single-process emulation of the item shards with oracle arithmetic (as tests/test_dist_cpu.py does)."""
    from test_dist_cpu import epoch_spec_blocks
    shards = [shard_problem(r, m, n, d, k) for r in range(nranks)]
    P = shards[0][1].copy()
    Qs = [sh[2].copy() for sh in shards]
    etot = float(sum(sh[0]['ev_ptr'][-1] for sh in shards))
    for epoch in range(epochs):
        blocks = [[] for _ in range(nranks)]
        for r in range(nranks):
            epoch_spec_blocks(orc, blocks[r], r, shards[r][0], P.copy(), Qs[r], 31, epoch, W, 0.05, 0.01, 0.01, etot, nranks)
        for per_rank in zip(*blocks):
            u0, u1 = per_rank[0][0], per_rank[0][1]
            total = per_rank[0][2].copy()
            for blk in per_rank[1:]:
                total += blk[2]                                   # rank order: the order the host-staged sum adds in
            P[u0:u1] += total
    return P, Qs, etot


def _two_shard_reference(orc, m, n, d, k, W, epochs):
    return _shard_reference(orc, m, n, d, k, W, epochs, 2)


def _spawn_two_ranks(tmp_path, script, args, local_ranks):
    _spawn_ranks(tmp_path, script, args, local_ranks)


def _spawn_ranks(tmp_path, script, args, local_ranks):
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    world = len(local_ranks)
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(local_ranks[r]), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, 'tests', 'helpers', script), str(tmp_path)] + [str(x) for x in args],
                                      cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=600)
        assert p.returncode == 0, out.decode()[-3000:]


def test_two_rank_epoch_on_one_gpu_through_the_test_seam(tmp_path, orc):
    """Two processes on device 0, item shards 0 and 1, the real yue_bpr_epoch on a 2-rank "communicator" whose collective is
    the test seam (host-staged sum over the control plane).  Several groups of user blocks per epoch (m * k * 4 = 15 MB against
    8 MB per group), so the second stream's reduce + apply really runs beside the next group's rounds.  Replicated user factors
    must come out bit-identical on both ranks and within 1e-5 of the two-shard spec."""
    import os
    from yue_amd.dist import epoch_block_plan
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, 'yue_amd', 'csrc', 'libyue_hip_seam.so')):
        pytest.fail('yue_amd/csrc/libyue_hip_seam.so missing: run __graft_entry__.build() (make -C yue_amd/csrc test-seam)')
    m, n, d, k, W, epochs = 30000, 4000, 12, 128, 16384, 2
    _spawn_two_ranks(tmp_path, 'seam_rank_main.py', (m, n, d, k, W, epochs), (0, 0))
    r0, r1 = np.load(tmp_path / 'seam_rank0.npz'), np.load(tmp_path / 'seam_rank1.npz')
    assert np.array_equal(r0['P'], r1['P'])                       # replicated user factors stay bit-identical
    assert r0['nll_total'] == r1['nll_total'] and abs(r0['nll_total'] - (r0['nll'] + r1['nll'])) <= 1e-9 * abs(r0['nll_total'])
    P, Qs, etot = _two_shard_reference(orc, m, n, d, k, W, epochs)
    assert rel_err(r0['P'], P) < 1e-5 and rel_err(r0['Q'], Qs[0]) < 1e-5 and rel_err(r1['Q'], Qs[1]) < 1e-5
    # the library reduced exactly the groups the plan announces: every user row once per epoch
    plan = epoch_block_plan(m, k, W, etot, 2)
    assert len(plan['groups']) > 1
    for r in (r0, r1):
        assert int(r['collectives']) == len(plan['groups']) and int(r['nranks']) == 2
        assert float(r['allreduce_bytes']) == 4.0 * m * k and int(r['elements']) == epochs * m * k
        # the groups' all-reduces and applies run on the second stream beside the next group's rounds: the compute stream waits
        # for that stream exactly once per epoch, behind the last group (never between two groups)
        assert int(r['compute_waits']) == 1 and int(r['group_mb']) == 8


def test_four_rank_epoch_on_one_gpu_through_the_test_seam(tmp_path, orc):
    """Four processes on device 0 (item shards 0..3), the round size left to the library (yue_default_round_events is a
    collective on a communicator: all ranks must arrive at the same value), k = 64, three epochs."""
    import os
    from yue_amd.dist import epoch_block_plan
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, 'yue_amd', 'csrc', 'libyue_hip_seam.so')):
        pytest.fail('yue_amd/csrc/libyue_hip_seam.so missing: run __graft_entry__.build() (make -C yue_amd/csrc test-seam)')
    m, n, d, k, epochs, nranks = 60000, 3000, 5, 64, 3, 4
    _spawn_ranks(tmp_path, 'seam_rank_main.py', (m, n, d, k, 0, epochs), (0,) * nranks)
    rs = [np.load(tmp_path / ('seam_rank%d.npz' % r)) for r in range(nranks)]
    W = int(rs[0]['round_events'])
    assert W > 0 and all(int(r['round_events']) == W for r in rs)
    for r in rs[1:]:
        assert np.array_equal(rs[0]['P'], r['P'])                 # replicated user factors stay bit-identical
        assert r['nll_total'] == rs[0]['nll_total']
    assert abs(rs[0]['nll_total'] - sum(float(r['nll']) for r in rs)) <= 1e-9 * abs(rs[0]['nll_total'])
    P, Qs, etot = _shard_reference(orc, m, n, d, k, W, epochs, nranks)
    assert rel_err(rs[0]['P'], P) < 1e-5
    for r in range(nranks):
        assert rel_err(rs[r]['Q'], Qs[r]) < 1e-5
    plan = epoch_block_plan(m, k, W, etot, nranks)
    for r in rs:
        assert int(r['collectives']) == len(plan['groups']) and int(r['nranks']) == nranks
        assert float(r['allreduce_bytes']) == 4.0 * m * k


def test_bench_multi_rank_flow_rehearsed_on_one_gpu(tmp_path):
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one process per rank), rehearsed on ONE GPU: both ranks
    on device 0, the collective through the test seam (YUE_BENCH_SEAM).  Checks this file's multi-rank flow -- rank / world from the
    launcher's environment, the control plane's barriers and max, the default workload for N > 1, the `comm` object, ONE JSON line
    from rank 0 -- not the interconnect."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, 'yue_amd', 'csrc', 'libyue_hip_seam.so')):
        pytest.fail('yue_amd/csrc/libyue_hip_seam.so missing: run __graft_entry__.build() (make -C yue_amd/csrc test-seam)')
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--workload', 'tiny']
    res = subprocess.run(cmd, cwd=root, env=dict(os.environ, YUE_BENCH_SEAM='1', OMP_NUM_THREADS='1', HSA_ENABLE_IPC_MODE_LEGACY='0'),
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert res.returncode == 0, res.stderr.decode()[-3000:]
    lines = [ln for ln in res.stdout.decode().splitlines() if ln.startswith('{')]
    assert len(lines) == 1, res.stdout.decode()[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 2 and out['warmup'] == 1 and out['value'] > 0 and out['scaling'] == 'weak'
    assert 'REHEARSAL' in out['data'] and 'cpu_baseline' not in out
    assert out['comm']['collectives_per_epoch'] >= 1 and out['comm']['nranks_rccl'] == 2
    assert out['comm']['allreduce_bytes_per_epoch_per_rank'] == 4.0 * 20000 * 128
    assert out['roofline']['frac'] > 0 and out['config']['parallelism'].startswith('items sharded x2')


def _device_count():
    import ctypes
    try:
        hip = ctypes.CDLL('libamdhip64.so')
        n = ctypes.c_int(0)
        return n.value if hip.hipGetDeviceCount(ctypes.byref(n)) == 0 else 0
    except OSError:
        return 0


def test_two_rank_product_epoch(tmp_path, orc):
    # two processes, one GPU each, the library's RCCL all-reduce between them (never run on the 1-GPU test boxes:
    # there it is skipped and the RCCL path stays verified for one rank only)
    if _device_count() < 2:
        pytest.skip('needs two GPUs')
    m, n, d, k, W, epochs = 3000, 2000, 25, 128, 4096, 2
    _spawn_two_ranks(tmp_path, 'gpu_rank_main.py', (m, n, d, k, W, epochs), (0, 1))
    r0, r1 = np.load(tmp_path / 'gpu_rank0.npz'), np.load(tmp_path / 'gpu_rank1.npz')
    assert np.array_equal(r0['P'], r1['P'])                       # replicated user factors stay bit-identical
    assert r0['nll_total'] == r1['nll_total'] and abs(r0['nll_total'] - (r0['nll'] + r1['nll'])) <= 1e-9 * abs(r0['nll_total'])
    P, Qs, _ = _two_shard_reference(orc, m, n, d, k, W, epochs)
    assert rel_err(r0['P'], P) < 1e-5 and rel_err(r0['Q'], Qs[0]) < 1e-5 and rel_err(r1['Q'], Qs[1]) < 1e-5


def test_multi_gpu_knobs_do_not_change_the_epoch(orc):
    """The two knobs kept for a real multi-GPU node -- comm_group_mb (MB of user-factor differences per all-reduce) and
    round_cus_reserved (CUs the compute stream leaves to RCCL's kernels: the stream is re-created with a CU mask) -- on a
    1-rank communicator: same factors as with the defaults, more / fewer collectives as asked."""
    from yue_amd._shim import Device, comm_unique_id
    from yue_amd import synth
    m, n, d, k, W = 40000, 3000, 10, 128, 16384
    data = synth.make_arrays(m, n, d, seed=3)
    P0, Q0 = synth.init_factors(m, n, k, 4)
    res = []
    for group_mb, reserved in ((8, 0), (2, 0), (64, 16)):
        dev = Device(0, raise_errors=True)
        try:
            dev.set_option('comm_group_mb', group_mb)
            dev.set_option('round_cus_reserved', reserved)
            dev.set_factors(P0, Q0)
            dev.set_interactions(data['indptr'], data['indices'], data['ev_ptr'], data['ev_i'])
            dev.comm_init(comm_unique_id(), 0, 1)
            nll = dev.bpr_epoch(7, 0, W, 0.02, 0.01, 0.01)[0]
            res.append((nll, dev.comm_stats()['collectives']) + dev.get_factors())
            assert dev.get_option('comm_last_compute_waits') == 1
        finally:
            dev.close()
    assert res[1][1] > res[0][1] > res[2][1] >= 1                  # 20 MB of user rows: 2 MB groups > 8 MB groups > one group
    for r in res[1:]:
        # (the same arithmetic; only the order of float-atomic sums and of the loss partials differs from run to run)
        assert abs(r[0] - res[0][0]) <= 1e-9 * abs(res[0][0]) and rel_err(r[2], res[0][2]) < 1e-6 and rel_err(r[3], res[0][3]) < 1e-6
