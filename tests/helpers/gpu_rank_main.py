"""Entry point of one GPU rank of tests/test_gpu_multi.py::test_two_rank_product_epoch: the PRODUCT path
(libyue_hip.so + RCCL) on device LOCAL_RANK, item shard RANK; no GPU call happens before this process starts."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

from yue_amd._shim import Device                       # noqa: E402
from yue_amd.dist import ControlPlane, attach_device   # noqa: E402
from sharded_spec import shard_problem                 # noqa: E402


def main():
    out_dir, m, n_local, d, k, W, epochs = sys.argv[1], *[int(x) for x in sys.argv[2:8]]
    cp = ControlPlane()
    data, P0, Q0 = shard_problem(cp.rank, m, n_local, d, k)
    dev = Device(cp.local_rank, raise_errors=True)
    dev.set_factors(P0, Q0)
    dev.set_interactions(data['indptr'], data['indices'], data['ev_ptr'], data['ev_i'])
    attach_device(dev, cp)
    nll = 0.0
    for epoch in range(epochs):
        nll, sp, sq = dev.bpr_epoch(31, epoch, W, 0.05, 0.01, 0.01)
    tot = dev.allreduce_f64([nll])[0]
    P, Q = dev.get_factors()
    np.savez(os.path.join(out_dir, 'gpu_rank%d.npz' % cp.rank), P=P, Q=Q, nll=nll, nll_total=tot)
    cp.barrier()
    dev.close()
    cp.close()


if __name__ == '__main__':
    main()
