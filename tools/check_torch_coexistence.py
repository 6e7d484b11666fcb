"""Does libyue_hip.so work in a process that has torch (ROCm build, its own HIP / RCCL libraries) loaded first,
as in `bench.py --gpus N` where torch.distributed (gloo) is the control plane?  Run on a GPU box."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29517')
t = time.time()
import torch                                    # noqa: E402
import torch.distributed as dist                # noqa: E402
dist.init_process_group('gloo', rank=0, world_size=1)
print('torch %s imported in %.1f s, hip %s' % (torch.__version__, time.time() - t, torch.version.hip))
import numpy as np                              # noqa: E402
from yue_amd import synth                       # noqa: E402
from yue_amd._shim import Device, comm_unique_id   # noqa: E402
m, n, d, k = 20000, 5000, 20, 128
data = synth.make_arrays(m, n, d, seed=3)
P0, Q0 = synth.init_factors(m, n, k, 4)
dev = Device(0, raise_errors=True)
dev.set_factors(P0, Q0)
dev.set_interactions(data['indptr'], data['indices'], data['ev_ptr'], data['ev_i'])
a = dev.bpr_epoch(1, 0, 8192, 0.02, 0.01, 0.01)
dev.comm_init(comm_unique_id(), 0, 1)
dev.set_factors(P0, Q0)
b = dev.bpr_epoch(1, 0, 8192, 0.02, 0.01, 0.01)
print('epoch without / with a 1-rank communicator:', a[0], b[0])
assert np.isfinite(a[0]) and abs(a[0] - b[0]) < 1e-3 * abs(a[0])
loaded = sorted({ln.split()[-1] for ln in open('/proc/self/maps') if ('libamdhip64' in ln or 'librccl' in ln or 'libyue_hip' in ln)})
print('\n'.join(loaded))
dev.close()
dist.destroy_process_group()
print('ok')
