"""Data-parallel training: one process per GPU, gradients summed with RCCL over xGMI.

The reference's multi-GPU mode is in-graph tower parallelism (mrcnn/parallel_model.py:54-104): the
batch is split across GPU_COUNT towers that share variables and the scalar losses are averaged.  Here
each rank runs the whole step on its IMAGES_PER_GPU images and the flat gradient buffer is all-reduced
(sum) and divided by the world size inside the optimiser kernel -- the same mean-of-per-tower-means.
Reduction is overlapped with the backward pass: the engine reports contiguous gradient ranges as soon
as they are final (heads+FPN first, then res5..res2, then the stem and BatchNorm block) and each range
is reduced on a side stream while the remaining backward kernels keep the compute stream busy.
xGMI is point-to-point (7 links/GPU), so few large messages beat many small ones: 7 ranges of 4-100 MB.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Initialise torch.distributed from the torchrun environment; returns (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:      # MRCNN_DIST_BACKEND=gloo: rehearse the multi-rank path where RCCL cannot run (ranks sharing a GPU)
            backend = os.environ.get("MRCNN_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("MRCNN_FORCE_DEVICE", local_rank)))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


class GradReducer(object):
    """Sums a flat gradient tensor over ranks, range by range, on a side stream."""

    def __init__(self, flat_grads, world_size):
        self.g = flat_grads
        self.world = world_size
        self.cuda = flat_grads.is_cuda
        self.stream = torch.cuda.Stream(device=flat_grads.device) if (self.cuda and world_size > 1) else None
        self.pending = []

    def ready(self, start, end):
        """Called by the engine when grads[start:end] are final."""
        if self.world <= 1 or end <= start:
            return
        view = self.g[start:end]
        if self.stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.g.device))
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ev)
                self.pending.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True))
        else:
            self.pending.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True))

    def finish(self):
        """Make the compute stream wait for every outstanding reduction."""
        for w in self.pending:
            w.wait()
        self.pending = []
        if self.stream is not None:
            torch.cuda.current_stream(self.g.device).wait_stream(self.stream)


def allreduce_mean_scalars(t, world_size):
    """5-float loss vector averaged over ranks (logging only)."""
    if world_size > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t /= world_size
    return t
