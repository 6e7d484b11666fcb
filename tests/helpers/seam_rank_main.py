"""Entry point of one rank of tests/test_gpu_multi.py::test_two_rank_epoch_on_one_gpu_through_the_test_seam: yue_bpr_epoch's
communicator path (identical user blocks on every rank, groups of blocks reduced on the second stream, k_apply_range) on
device 0, item shard RANK, with the collective behind libyue_hip_seam.so's test seam: a host-staged sum over the
standard-library control plane (yue_amd/dist.py).  Test infrastructure: the product library has no such entry point."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from yue_amd import _shim                              # noqa: E402
_shim.LIB_PATH = os.path.join(ROOT, 'yue_amd', 'csrc', 'libyue_hip_seam.so')
from yue_amd._shim import Device                       # noqa: E402
from yue_amd.dist import ControlPlane                  # noqa: E402
from sharded_spec import shard_problem                 # noqa: E402

REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_void_p)


def main():
    out_dir, m, n_local, d, k, W, epochs = sys.argv[1], *[int(x) for x in sys.argv[2:8]]
    cp = ControlPlane()
    calls = {'f32': 0, 'f64': 0, 'elements': 0}

    def reduce(host, count, dtype, user):
        try:
            ct = C.c_double if dtype else C.c_float
            arr = np.ctypeslib.as_array((ct * count).from_address(host))
            cp.allreduce_sum(arr)
            calls['f64' if dtype else 'f32'] += 1
            calls['elements'] += 0 if dtype else count
            return 0
        except Exception as exc:                                   # never let an exception cross the C boundary
            print('seam reduce failed:', exc)
            return 1
    cb = REDUCE_FN(reduce)
    data, P0, Q0 = shard_problem(cp.rank, m, n_local, d, k)
    dev = Device(0, raise_errors=True)
    dev.set_factors(P0, Q0)
    dev.set_interactions(data['indptr'], data['indices'], data['ev_ptr'], data['ev_i'])
    lib = dev._lib
    lib.yue_seam_init.restype = C.c_int
    assert lib.yue_seam_init(dev._ctx, C.c_int(cp.rank), C.c_int(cp.world), cb, None) == 0
    if W == 0:
        W = dev.default_round_events()                             # collective (through the seam): the same value on every rank
    nll = 0.0
    for epoch in range(epochs):
        nll, sp, sq = dev.bpr_epoch(31, epoch, W, 0.05, 0.01, 0.01)
    stats = dev.comm_stats()
    waits = dev.get_option('comm_last_compute_waits')      # the compute stream waits for the collective stream once per epoch: behind its last group
    tot = dev.allreduce_f64([nll])[0]
    P, Q = dev.get_factors()
    np.savez(os.path.join(out_dir, 'seam_rank%d.npz' % cp.rank), P=P, Q=Q, nll=nll, nll_total=tot, f32_calls=calls['f32'], elements=calls['elements'],
             collectives=stats['collectives'], allreduce_bytes=stats['allreduce_bytes'], nranks=stats['nranks'], round_events=W, compute_waits=waits,
             group_mb=dev.get_option('comm_group_mb'))
    cp.barrier()
    dev.close()
    cp.close()


if __name__ == '__main__':
    main()
