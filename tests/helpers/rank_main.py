"""Entry point of one CPU rank (launched by tests/test_dist_cpu.py, directly or through torch.distributed.run):
the control plane of yue_amd/dist.py (standard-library TCP star) + the sharded S-round epoch as an executable
spec with oracle arithmetic."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

import oracle                                    # noqa: E402
from yue_amd.dist import ControlPlane            # noqa: E402
from sharded_spec import epoch_spec, shard_problem   # noqa: E402


def main():
    out_dir = sys.argv[1]
    cp = ControlPlane()
    assert 'torch' not in sys.modules            # the control plane must not pull torch into a rank
    orc = oracle.Oracle()
    m, n_local, d, k = 300, 200, 12, 16
    data, P, Q = shard_problem(cp.rank, m, n_local, d, k)
    token = cp.broadcast_bytes(b'id-from-rank-0' * 9 if cp.rank == 0 else None)
    assert token == b'id-from-rank-0' * 9
    last = cp.broadcast_bytes(b'from-the-last-rank' if cp.rank == cp.world - 1 else None, src=cp.world - 1)
    assert last == b'from-the-last-rank'
    events_total = float(cp.allreduce_sum(np.array([float(data['ev_ptr'][-1])], np.float64))[0])
    nll = 0.0
    for epoch in range(2):
        nll = epoch_spec(orc, cp.allreduce_sum, cp.world, cp.rank, data, P, Q, 77, epoch, 256, 0.05, 0.01, 0.01, events_total)
    cp.barrier()
    slowest = cp.reduce_max(float(cp.rank))
    assert slowest == cp.world - 1
    np.savez(os.path.join(out_dir, 'rank%d.npz' % cp.rank), P=P, Q=Q, nll=nll, events_total=events_total)
    cp.close()


if __name__ == '__main__':
    main()
