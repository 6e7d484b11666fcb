"""GPU: the communicator path of yue_bpr_epoch (user-aligned blocks, RCCL all-reduce of the block's
user-factor differences in place, range apply) with a 1-rank communicator on the one GPU we have,
against the executable spec that tests/test_dist_cpu.py runs on two gloo ranks."""
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


def test_library_works_after_torch_was_loaded():
    # bench.py --gpus N imports torch.distributed (gloo control plane) before the library; the ROCm build of
    # torch brings its own libamdhip64 / librccl, which the library then binds to.  Fresh interpreter.
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29531')
    res = subprocess.run([sys.executable, os.path.join(root, 'tools', 'check_torch_coexistence.py')], stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, timeout=600, env=env)
    out = res.stdout.decode()
    assert res.returncode == 0 and out.strip().endswith('ok'), out[-2000:]
