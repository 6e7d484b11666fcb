"""GPU: the communicator path of yue_bpr_epoch (user-aligned blocks, RCCL all-reduce of the block's
user-factor differences in place, range apply) with a 1-rank communicator on the one GPU we have,
against the executable spec that tests/test_dist_cpu.py runs on two CPU ranks; and the real thing --
two processes, two GPUs, RCCL -- wherever the box has two devices (skipped otherwise)."""
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
    import os
    import socket
    import subprocess
    import sys
    if _device_count() < 2:
        pytest.skip('needs two GPUs')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    m, n, d, k, W, epochs = 3000, 2000, 25, 128, 4096, 2
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, 'tests', 'helpers', 'gpu_rank_main.py'), str(tmp_path)] + [str(x) for x in (m, n, d, k, W, epochs)],
                                      cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=600)
        assert p.returncode == 0, out.decode()[-3000:]
    r0, r1 = np.load(tmp_path / 'gpu_rank0.npz'), np.load(tmp_path / 'gpu_rank1.npz')
    assert np.array_equal(r0['P'], r1['P'])                       # replicated user factors stay bit-identical
    assert r0['nll_total'] == r1['nll_total'] and abs(r0['nll_total'] - (r0['nll'] + r1['nll'])) <= 1e-9 * abs(r0['nll_total'])
    # single-process emulation of the two shards with oracle arithmetic (as tests/test_dist_cpu.py does)
    from test_dist_cpu import epoch_spec_blocks
    shards = [shard_problem(r, m, n, d, k) for r in range(2)]
    P = shards[0][1].copy()
    Qs = [sh[2].copy() for sh in shards]
    etot = float(sum(sh[0]['ev_ptr'][-1] for sh in shards))
    for epoch in range(epochs):
        blocks = [[], []]
        for r in range(2):
            epoch_spec_blocks(orc, blocks[r], r, shards[r][0], P.copy(), Qs[r], 31, epoch, W, 0.05, 0.01, 0.01, etot)
        for (u0, u1, b0), (_, _, b1) in zip(blocks[0], blocks[1]):
            P[u0:u1] += b0 + b1
    assert rel_err(r0['P'], P) < 1e-5 and rel_err(r0['Q'], Qs[0]) < 1e-5 and rel_err(r1['Q'], Qs[1]) < 1e-5
