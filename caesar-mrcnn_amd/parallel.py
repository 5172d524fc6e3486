"""Data-parallel training: one process per GPU, gradients summed with RCCL over xGMI.

The reference's multi-GPU mode is in-graph tower parallelism (mrcnn/parallel_model.py:54-104): the
batch is split across GPU_COUNT towers that share variables and the scalar losses are averaged.  Here
each rank runs the whole step on its IMAGES_PER_GPU images and the flat gradient buffer is all-reduced
(sum) and divided by the world size inside the optimiser kernel -- the same mean-of-per-tower-means.
Reduction is overlapped with the backward pass: the engine reports contiguous gradient ranges as soon
as they are final (heads+FPN first, then res5..res2, then the stem and BatchNorm block) and each range
is reduced on a side stream while the remaining backward kernels keep the compute stream busy.
xGMI is point-to-point (7 links/GPU), so few large messages beat many small ones: 7 ranges of 4-100 MB
by default; ``MRCNN_ALLREDUCE_MAX_MB`` cuts ranges into pieces of at most that size.

Transport.  On GPUs the exchange goes through the C-ABI (include/mrcnn_hip.h: mrcnn_allreduce_init /
mrcnn_allreduce_grad, RCCL bound at run time); torch.distributed only carries the 128-byte rendezvous id
and the logging scalars.  ``MRCNN_ALLREDUCE=direct`` selects the reduce-scatter + all-gather written as
grouped point-to-point transfers (one per xGMI link), ``MRCNN_ALLREDUCE=torch`` (and every run without a
GPU, i.e. the gloo tests) the plain ``dist.all_reduce`` per range.
"""
import ctypes as C
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


def split_range(start, end, max_floats):
    """[start, end) in pieces of at most max_floats (multiples of 64 floats, the layout's granule); None = one piece."""
    if not max_floats or end - start <= max_floats:
        return [(start, end)]
    step = max(64, int(max_floats) // 64 * 64)
    return [(a, min(a + step, end)) for a in range(start, end, step)]


class RcclComm(object):
    """The C-ABI communicator (mrcnn_allreduce_*): created collectively by all ranks; the 128-byte id travels over
    torch.distributed (whatever backend it runs on) or, for a single rank, nowhere."""

    def __init__(self, rank, world, device):
        from . import _hip
        self.lib = _hip.lib()
        self.rank, self.world = rank, world
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        rc = self.lib.mrcnn_allreduce_load(path.encode() if os.path.exists(path) else None)
        if rc != 0:
            raise _hip.HipPathError("RCCL could not be loaded: %s" % self.lib.mrcnn_allreduce_last_error().decode())
        ident = (C.c_ubyte * 128)()
        if rank == 0:
            _hip.check(self.lib.mrcnn_allreduce_unique_id(ident), "mrcnn_allreduce_unique_id")
        if world > 1:
            box = [bytes(ident)]
            dist.broadcast_object_list(box, src=0)
            ident = (C.c_ubyte * 128).from_buffer_copy(box[0])
        torch.cuda.set_device(device)
        self.handle = C.c_void_p()
        rc = self.lib.mrcnn_allreduce_init(C.byref(self.handle), ident, rank, world)
        if rc != 0:
            raise _hip.HipPathError("mrcnn_allreduce_init failed (%d): %s" % (rc, self.lib.mrcnn_allreduce_last_error().decode()))

    def reduce(self, flat, start, end, algo, scratch, stream):
        from . import _hip
        fn = self.lib.mrcnn_allreduce_grad
        args = (self.handle, flat.data_ptr(), start, end, algo, scratch.data_ptr() if scratch is not None else None,
                scratch.numel() * 4 if scratch is not None else 0, stream)
        rc = fn(*args)
        if rc != 0:
            raise _hip.HipPathError("mrcnn_allreduce_grad failed (%d): %s" % (rc, self.lib.mrcnn_allreduce_last_error().decode()))
        if _hip._tape is not None:          # part of a recorded step (engine.step_taped): re-issued with the step's launches
            _hip._tape.append((fn, args))

    def self_test(self, device, algo, scratch, stream):
        """Collective, once per communicator: a range whose sum over ranks is known in closed form (integer-valued floats, so
        every summation order gives the same bits) goes through exactly the call the training step will make; any rank that
        does not see the expected sum raises.  The first multi-GPU run of a transport is then also its first check."""
        from . import _hip
        n = 64 * self.world * 5 + 37                        # not a multiple of 64 * world: short last chunk
        base = torch.arange(n, dtype=torch.float32, device=device) % 251.0
        buf = torch.zeros(n + 128, dtype=torch.float32, device=device)
        buf[64:64 + n] = base * float(self.rank + 1)
        s = torch.cuda.Stream(device=device) if stream is None else stream
        s.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(s):
            self.reduce(buf, 64, 64 + n, algo, scratch, s.cuda_stream)
        torch.cuda.current_stream(device).wait_stream(s)
        want = base * float(self.world * (self.world + 1) // 2)
        ok = bool(torch.equal(buf[64:64 + n], want)) and float(buf[:64].abs().sum()) == 0.0 and float(buf[64 + n:].abs().sum()) == 0.0
        if not ok:
            bad = int((buf[64:64 + n] != want).sum())
            raise _hip.HipPathError("gradient exchange self-test failed on rank %d of %d (algo %d): %d of %d elements differ from "
                                    "the expected sum -- refusing to train on a broken transport; MRCNN_ALLREDUCE=torch selects "
                                    "torch.distributed's all_reduce" % (self.rank, self.world, algo, bad, n))

    def close(self):
        if self.handle:
            self.lib.mrcnn_allreduce_destroy(self.handle)
            self.handle = C.c_void_p()


def _reraise(why):
    from . import _hip
    return _hip.HipPathError("gradient exchange set-up failed: %s" % why)


class GradReducer(object):
    """Sums a flat gradient tensor over ranks, range by range, on a side stream.

    ``timing=True`` brackets every exchange with HIP events on the exchange stream: ``pop_timing()`` then returns the
    milliseconds the exchanges of the finished steps took there and the bytes they moved (bench.py, N > 1)."""

    def __init__(self, flat_grads, world_size, rank=None, mode=None, timing=False, max_mb=None, force=False):
        """force: build the transport even for a world of one rank (the sum is then the identity; tests run the whole path --
        hooks, side stream, C-ABI call, launch tape -- on the single GPU of this pool)."""
        self.g = flat_grads
        self.world = world_size
        self.cuda = flat_grads.is_cuda
        self.active = world_size > 1 or bool(force)
        self.stream = torch.cuda.Stream(device=flat_grads.device) if (self.cuda and self.active) else None
        self.pending = []
        self.timing = timing
        self._events, self.bytes_moved, self.range_log = [], 0, []
        mode = mode or os.environ.get("MRCNN_ALLREDUCE") or ("rccl" if self.cuda else "torch")
        if dist.is_initialized() and dist.get_backend() == "gloo":
            mode = "torch"                                  # ranks sharing one GPU (rehearsals) cannot form an RCCL communicator
        self.mode = mode if self.active else "none"
        mb = max_mb if max_mb is not None else os.environ.get("MRCNN_ALLREDUCE_MAX_MB")
        self.max_floats = int(float(mb) * (1 << 20) / 4) if mb else None
        self.comm = self.scratch = None
        if self.mode in ("rccl", "direct"):
            rank = (dist.get_rank() if dist.is_initialized() else 0) if rank is None else rank
            why = None
            try:
                self.comm = RcclComm(rank, world_size, flat_grads.device)
                self.algo = 1 if self.mode == "direct" else 0
                if self.algo == 1:
                    longest = self.max_floats or flat_grads.numel()
                    nbytes = self.comm.lib.mrcnn_allreduce_scratch(world_size, min(longest, flat_grads.numel()), 1)
                    self.scratch = torch.empty(max(nbytes // 4, 64), dtype=torch.float32, device=flat_grads.device)
                # neither form has run on more than one GPU of this pool: the first thing a communicator does is prove itself
                self.comm.self_test(flat_grads.device, self.algo, self.scratch, self.stream)
            except Exception as e:                      # decided TOGETHER below: one rank must not leave the others in a collective
                why = repr(e)
            if world_size > 1 and dist.is_initialized():
                ok = torch.tensor([0.0 if why else 1.0], device=flat_grads.device)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if float(ok.item()) == 0.0:
                    # still RCCL over xGMI (torch.distributed's own communicator), never a host path: the step keeps its overlap
                    if rank == 0:
                        import sys
                        sys.stderr.write("[mrcnn] gradient exchange through the C-ABI (%s) failed its start-up check on some rank%s; "
                                         "every rank falls back to torch.distributed.all_reduce\n" % (self.mode, ": " + why if why else ""))
                    if self.comm is not None:
                        try:
                            self.comm.close()
                        except Exception:
                            pass
                    self.comm = self.scratch = None
                    self.mode = "torch"
            elif why:
                raise _reraise(why)

    def ready(self, start, end):
        """Called by the engine when grads[start:end] are final.  Stream hand-offs go through _hip.ev_record / ev_wait and the
        exchange calls append themselves to an open launch tape, so a recorded step (engine.step_taped) re-issues the
        exchange with the rest of its launches."""
        from . import _hip
        if not self.active or end <= start:
            return
        pieces = split_range(start, end, self.max_floats)
        if self.stream is not None:
            ev = _hip.ev_record(torch.cuda.current_stream(self.g.device))
            with torch.cuda.stream(self.stream):
                _hip.ev_wait(self.stream, ev)
                if self.timing:
                    e0 = torch.cuda.Event(enable_timing=True)
                    e0.record(self.stream)
                for a, b in pieces:
                    self._one(a, b)
                if self.timing:
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record(self.stream)
                    self._events.append((e0, e1))
        else:
            for a, b in pieces:
                self._one(a, b)
        if self.timing:                                     # only a caller that drains it (pop_timing) makes it grow
            self.bytes_moved += (end - start) * 4
            self.range_log.append((end - start) * 4)

    def _torch_allreduce(self, a, b):
        if self.stream is not None:
            with torch.cuda.stream(self.stream):
                self.pending.append(dist.all_reduce(self.g[a:b], op=dist.ReduceOp.SUM, async_op=True))
        else:
            self.pending.append(dist.all_reduce(self.g[a:b], op=dist.ReduceOp.SUM, async_op=True))

    def _one(self, a, b):
        from . import _hip
        if self.comm is not None:
            self.comm.reduce(self.g, a, b, self.algo, self.scratch, self.stream.cuda_stream)
        elif self.world > 1:
            self._torch_allreduce(a, b)
            if _hip._tape is not None:
                _hip._tape.append((self._torch_allreduce, (a, b)))

    def _finish_pending(self):
        for w in self.pending:
            w.wait()
        self.pending = []

    def finish(self):
        """Make the compute stream wait for every outstanding reduction."""
        from . import _hip
        self._finish_pending()
        if _hip._tape is not None and self.comm is None and self.world > 1:
            _hip._tape.append((self._finish_pending, ()))
        if self.stream is not None:
            _hip.stream_wait(torch.cuda.current_stream(self.g.device), self.stream)

    def pop_timing(self):
        """(ms on the exchange stream, bytes reduced, per-range byte sizes of the last step) since the last call;
        synchronises."""
        torch.cuda.synchronize(self.g.device)
        ms = sum(e0.elapsed_time(e1) for e0, e1 in self._events)
        out = (ms, self.bytes_moved, list(self.range_log))
        self._events, self.bytes_moved, self.range_log = [], 0, []
        return out

    def close(self):
        if self.comm is not None:
            self.comm.close()
            self.comm = None


def allreduce_mean_scalars(t, world_size):
    """5-float loss vector averaged over ranks (logging only)."""
    if world_size > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t /= world_size
    return t
