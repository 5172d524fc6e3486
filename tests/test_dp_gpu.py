"""Data-parallel training step with 2 ranks.  The GPU box has one MI355X, so both ranks drive cuda:0
and exchange gradients over gloo (CUDA tensors): this exercises exactly the code RCCL runs on a node --
engine gradient-ready hooks, the side-stream GradReducer, grad_prepare(1/world) and the SGD update --
and checks the result against gradients computed rank by rank in one process."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "oracle")); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch, torch.distributed as dist
import test_engine_gpu as T
from caesar_mrcnn_amd.model import MaskRCNN
from caesar_mrcnn_amd.parallel import GradReducer, init_distributed
rank, _, world = init_distributed(backend="gloo")
dev = torch.device("cuda:0")
cfg = T._small_cfg("custom", 128)
w = T._weights(cfg, 23)
model = MaskRCNN("training", cfg, "/tmp/mrcnn_logs", device=dev, weights=w)
model.compile(0.01, 0.9)
inputs, keys = T._train_inputs(cfg, 2, 31 + rank)          # each rank its own images
red = GradReducer(model.engine.grads, world)
losses = model.train_on_batch(inputs, rand_keys=keys, reducer=red, world_size=world)
torch.cuda.synchronize()
np.save(%(out)r + "/params_rank%%d.npy" %% rank, model.engine.params.cpu().numpy())
np.save(%(out)r + "/losses_rank%%d.npy" %% rank, losses.cpu().numpy())
dist.barrier(); dist.destroy_process_group()
"""


def test_two_rank_step_matches_manual_average(dev, tmp_path):
    import test_engine_gpu as T
    from caesar_mrcnn_amd.model import MaskRCNN
    script = tmp_path / "dp_worker.py"
    script.write_text(_WORKER % {"root": ROOT, "out": str(tmp_path)})
    port = 29700 + (os.getpid() % 200)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=280)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    p0 = np.load(str(tmp_path / "params_rank0.npy")); p1 = np.load(str(tmp_path / "params_rank1.npy"))
    assert np.array_equal(p0, p1), "ranks diverged after one data-parallel step"
    # reference: the two per-rank gradients computed here, summed, then one update with world_size=2
    cfg = T._small_cfg("custom", 128)
    w = T._weights(cfg, 23)
    model = MaskRCNN("training", cfg, "/tmp/mrcnn_logs", device=dev, weights=w)
    model.compile(0.01, 0.9)
    total = None
    for r in range(2):
        inputs, keys = T._train_inputs(cfg, 2, 31 + r)
        losses = model.train_on_batch(inputs, rand_keys=keys, apply=False)
        np.testing.assert_allclose(losses.cpu().numpy(), np.load(str(tmp_path / ("losses_rank%d.npy" % r))), rtol=1e-5)
        g = model.engine.grads.clone()
        total = g if total is None else total + g
    model.engine.grads.copy_(total)
    model.engine.apply_gradients(0.01, 0.9, world_size=2)
    torch.cuda.synchronize()
    ref = model.engine.params.cpu().numpy()
    scale = np.abs(ref).max()
    assert np.abs(ref - p0).max() <= 1e-5 * scale


def test_rccl_cabi_single_rank(dev):
    """The C-ABI RCCL binding (mrcnn_allreduce_load / _unique_id / _init / _grad / _destroy) end to end on the one GPU of
    this pool: a communicator of one rank, where the sum over ranks is the identity -- checks the run-time binding (the
    librccl PyTorch already loaded), the rendezvous id, the enqueue on a side stream and both algorithms' argument
    handling.  More than one rank needs more than one GPU (RCCL refuses two ranks on one device): the multi-rank
    arithmetic of the data-parallel step is covered over gloo above, the transport itself only on a real node."""
    from caesar_mrcnn_amd.parallel import GradReducer, RcclComm
    comm = RcclComm(0, 1, dev)
    g = torch.arange(70000, dtype=torch.float32, device=dev)
    ref = g.clone()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        comm.reduce(g, 0, 70000, 0, None, side.cuda_stream)
        comm.reduce(g, 128, 4096, 0, None, side.cuda_stream)
        comm.reduce(g, 0, 70000, 1, None, side.cuda_stream)        # direct form: nothing to exchange with one rank
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize()
    assert torch.equal(g, ref)
    assert comm.lib.mrcnn_allreduce_scratch(8, 1 << 20, 1) == 7 * (1 << 17) * 4
    assert comm.lib.mrcnn_allreduce_scratch(8, 1 << 20, 0) == 0
    comm.close()
    red = GradReducer(g, 1)                                         # world 1: no communicator, ready() is a no-op
    red.ready(0, 100)
    red.finish()
    assert red.mode == "none" and red.comm is None


@pytest.mark.parametrize("world", [2, 3, 8])
def test_direct_allreduce_simulated_ranks(dev, world):
    """The LOCAL half of the direct reduce-scatter + all-gather (chunk_of / direct_chunk / the scratch-slot indexing and
    allreduce_sum_chunks_kernel of csrc/allreduce.hip) with fabricated peers: `world` gradient buffers on the one GPU,
    mrcnn_allreduce_direct_simulate pairs every planned send with the receive its peer planned (equal lengths or it
    fails), moves the chunks with device copies and sums them with the real kernel.  Every "rank" must end with the
    rank-ordered float32 sum, bit for bit, and untouched floats outside the range -- for ranges that are not multiples of
    64 * world, ranges with empty trailing chunks and a one-float range."""
    import ctypes as C
    from caesar_mrcnn_amd import _hip
    L = _hip.lib()
    total = 70000
    gen = torch.Generator(device="cpu").manual_seed(world)
    for start, end in ((0, total), (64, 64 + 64 * world * 7), (128, 128 + 6411), (5, 6), (9, 9 + world - 1), (192, 192 + 64 * world + 1)):
        grads = [torch.randn(total, generator=gen).to(dev) for _ in range(world)]
        before = [g.clone() for g in grads]
        want = before[0][start:end].clone()
        for r in range(1, world):
            want = want + before[r][start:end]
        nbytes = L.mrcnn_allreduce_scratch(world, end - start, 1)
        scratch = [torch.full((max(nbytes // 4, 1),), float("nan"), device=dev) for _ in range(world)]
        gp = (C.c_void_p * world)(*[g.data_ptr() for g in grads])
        sp = (C.c_void_p * world)(*[t.data_ptr() for t in scratch])
        rc = L.mrcnn_allreduce_direct_simulate(gp, sp, nbytes, world, start, end, _hip.current_stream())
        assert rc == 0, L.mrcnn_allreduce_last_error()
        torch.cuda.synchronize()
        for r in range(world):
            assert torch.equal(grads[r][start:end], want), (world, start, end, r)
            assert torch.equal(grads[r][:start], before[r][:start]) and torch.equal(grads[r][end:], before[r][end:])
    assert L.mrcnn_allreduce_direct_simulate(gp, sp, 0, world, 0, total, _hip.current_stream()) == -3      # scratch too small


def test_taped_data_parallel_step_equals_eager(dev):
    """The gradient hooks are part of the launch tape (round 2 switched the tape off under data parallelism): a
    GradReducer over a one-rank RCCL communicator (force=True: the sum is the identity, the path is the real one --
    hand-off events, exchange stream, mrcnn_allreduce_grad through the C-ABI, the join) drives three taped steps; they
    must equal three eager steps with the same reducer, the recording must hold one exchange call per gradient range and
    the communicator's start-up self-test must have passed."""
    import test_engine_gpu as T
    from caesar_mrcnn_amd import _hip
    from caesar_mrcnn_amd.model import MaskRCNN
    from caesar_mrcnn_amd.parallel import GradReducer
    cfg = T._small_cfg("custom", 128)
    w = T._weights(cfg, 23)
    batches = [T._train_inputs(cfg, 2, 31), T._train_inputs(cfg, 2, 33)]
    out = {}
    for taped in (False, True):
        cfg.TRAIN_LAUNCH_TAPE = taped
        model = MaskRCNN("training", cfg, "/tmp/mrcnn_logs", device=dev, weights=w)
        model.compile(0.01, 0.9)
        red = GradReducer(model.engine.grads, 1, rank=0, mode="rccl", force=True)
        assert red.mode == "rccl" and red.comm is not None
        losses = []
        for s in range(3):
            inputs, keys = batches[s % 2]
            losses.append(model.train_on_batch(inputs, rand_keys=keys, reducer=red, world_size=1).cpu().numpy().copy())
        torch.cuda.synchronize()
        out[taped] = (np.stack(losses), model.engine.params.cpu().numpy().copy())
        if taped:
            assert len(model.engine._train_tapes) == 1
            tape = list(model.engine._train_tapes.values())[0][0]
            real = _hip.lib().mrcnn_allreduce_grad
            n_exchange = sum(1 for f, a in tape if f is real)
            assert n_exchange == len(model.engine.grad_ranges), (n_exchange, len(model.engine.grad_ranges))
        red.close()
    np.testing.assert_allclose(out[True][0], out[False][0], rtol=2e-4, atol=1e-5)
    assert np.abs(out[True][1] - out[False][1]).max() <= 2e-4 * np.abs(out[False][1]).max()
