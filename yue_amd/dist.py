"""Control plane for one-process-per-GPU runs (bench.py, multi-GPU training).

The data path is RCCL inside libyue_hip.so (yue_comm_init / yue_bpr_epoch).  This module only
ships the RCCL unique id, synchronises and reduces scalars between the ranks; it uses
torch.distributed with the gloo backend (CPU), which the launcher
(`python -m torch.distributed.run`) has already configured through RANK / WORLD_SIZE / MASTER_*.
"""
import math
import os


def env_rank():
    """(rank, world, local_rank) from the launcher's environment; (0, 1, 0) when run directly."""
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('LOCAL_RANK', '0')))


class ControlPlane(object):
    def __init__(self):
        self.rank, self.world, self.local_rank = env_rank()
        self._dist = None
        if self.world > 1:
            import torch.distributed as dist
            if not dist.is_initialized():
                dist.init_process_group('gloo', rank=self.rank, world_size=self.world)
            self._dist = dist

    def broadcast_bytes(self, payload, src=0):
        """payload is given on `src` (None elsewhere); every rank gets the bytes."""
        if self._dist is None:
            return payload
        box = [payload if self.rank == src else None]
        self._dist.broadcast_object_list(box, src=src)
        return box[0]

    def barrier(self):
        if self._dist is not None:
            self._dist.barrier()

    def reduce_max(self, x):
        if self._dist is None:
            return float(x)
        import torch
        t = torch.tensor([float(x)], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t[0])

    def allreduce_sum(self, array):
        """In-place float32/float64 numpy sum over ranks (CPU; used by the executable spec in tests)."""
        if self._dist is None:
            return array
        import torch
        t = torch.from_numpy(array)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return array

    def close(self):
        if self._dist is not None and self._dist.is_initialized():
            self._dist.destroy_process_group()
            self._dist = None


def user_block_width(round_events, events_total, m, world):
    """Users per round on a communicator: the same number on every rank, derived from job-wide
    counts only (mirror of the rule in csrc/yue_hip.hip: yue_bpr_epoch)."""
    per_user = float(events_total) / world / m
    return max(1, int(math.floor(round_events / max(per_user, 1e-9) + 0.5)))


def epoch_round_ptr(ev_ptr, round_events, events_total=None, world=1):
    """Round boundaries (event offsets) of yue_bpr_epoch for this rank's events: blocks of whole users,
    user_block_width users each (the last block takes the rest)."""
    m = len(ev_ptr) - 1
    total = int(ev_ptr[-1]) if events_total is None else events_total
    ub = user_block_width(round_events, total, m, world)
    return [int(ev_ptr[u]) for u in range(0, m, ub)] + [int(ev_ptr[-1])]


def attach_device(dev, cp):
    """Join `dev` (a yue_amd._shim.Device) to the job's RCCL communicator."""
    if cp.world > 1:
        from ._shim import comm_unique_id
        ident = cp.broadcast_bytes(comm_unique_id() if cp.rank == 0 else None)
        dev.comm_init(ident, cp.rank, cp.world)
