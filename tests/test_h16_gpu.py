"""16-bit matrix-core path (BASELINE configs[4], stage 1): float16 / bfloat16 convolutions of the ROI heads
against the float32 oracle evaluated on the SAME 16-bit-rounded operands.  The kernel multiplies exactly
(16-bit x 16-bit products are exact in float32) and accumulates in float32, so what is left is summation order
and the single output rounding: tolerance 2^-10 (f16) / 2^-7 (bf16) of the output magnitude, stated below."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import mrcnn_oracle as orc

pytestmark = pytest.mark.gpu

TOL = {torch.float16: 2.0 ** -10, torch.bfloat16: 2.0 ** -7}


def _ops():
    import caesar_mrcnn_amd  # noqa: F401
    from caesar_mrcnn_amd import ops
    return ops


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "the -m gpu tests need an MI355X"
    return torch.device("cuda:0")


H16_CASES = [
    # N, H, W, Cin, Cout, k, padding, act, bn
    (64, 14, 14, 256, 256, 3, "same", 1, True),      # mask-head conv: padded taps, several 256-row tiles
    (90, 13, 11, 128, 256, 3, "same", 0, False),     # ragged M (12870), odd H/W
    (5, 16, 16, 128, 128, 1, "valid", 1, True),      # 1x1, one N tile
    (300, 7, 7, 256, 1024, 7, "valid", 1, True),     # class-head FC as 7x7 VALID conv (49 taps, 64-bit tap mask)
    (7, 14, 14, 32, 512, 3, "same", 1, False),       # Ktot = 288 = 9 K-steps: ring prologue / drain with nk % 4 == 1, ragged M
    (2, 5, 5, 64, 256, 1, "valid", 0, True),         # 2 K-steps only: fewer steps than ring slots
    (9, 12, 12, 64, 512, 3, "same", 1, True),        # phased tile: 9 K-steps of 64 (odd: one zero tile), two N tiles, ragged M
]


@pytest.fixture(params=["small", "big", "phase", "slab", "wave128"])
def h16_tile(request):
    """The forward tilings on every eligible shape: MRCNN_H16_TILE is read per call (256 x 128 / 4 waves / double
    buffered; 256 x 256 / 8 waves / 4-stage ring with counted waits; 256 x 256 / 8 waves in two staggered groups, phased
    K-steps; 256 x 256 / 4 waves of 128 x 128 with software-pipelined operand reads and the epilogue through LDS -- the last
    three need Cout % 256 == 0, the phased one also Cin % 64 == 0, else the call takes the default)."""
    # "slab": the phased tile with the pixels of a channel chunk staged once and the taps as shifted reads (conv_fwd_h16q_kernel,
    # what eligible shapes take by default); "phase" keeps the per-tap staging on those shapes too
    os.environ["MRCNN_H16_TILE"] = "phase" if request.param == "slab" else request.param
    _ops().tuning_set("h16_slab", 1 if request.param == "slab" else 0)
    yield request.param
    _ops().tuning_set("h16_slab", -1)
    del os.environ["MRCNN_H16_TILE"]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("case", H16_CASES)
def test_conv_fwd_h16(dev, case, dtype, h16_tile):
    ops = _ops()
    N, H, W, Cin, Cout, k, padding, act, bn = case
    if h16_tile in ("big", "phase", "slab", "wave128") and (Cout % 256 or (h16_tile in ("phase", "slab") and Cin % 64)):
        pytest.skip("256 x 256 tile needs Cout % 256 == 0")
    if h16_tile == "slab" and not (k == 3 and padding == "same" and W <= 14 and Cin % 128 == 0):
        pytest.skip("slab staging: 3 x 3 on maps no wider than 14 pixels, an even number of 64-channel chunks")
    rng = np.random.default_rng(sum(case[:6]))
    x = torch.tensor(rng.standard_normal((N, H, W, Cin)).astype(np.float32)).to(dtype)
    w = torch.tensor((rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)).to(dtype)
    b = torch.tensor(rng.standard_normal(Cout).astype(np.float32) * 0.1)
    z_ref = orc.conv2d_nhwc(x.float(), w.float(), b, 1, padding)
    y_ref = z_ref
    scale = shift = None
    if bn:
        scale = torch.tensor(rng.uniform(0.5, 1.5, Cout).astype(np.float32))
        shift = torch.tensor(rng.uniform(-0.2, 0.2, Cout).astype(np.float32))
        y_ref = z_ref * scale + shift
    if act == 1:
        y_ref = torch.relu(y_ref)
    wf, wd = ops.weights_to_h16(w.float().to(dev), dtype)
    np.testing.assert_array_equal(wf.float().cpu().numpy(), w.float().permute(3, 0, 1, 2).reshape(Cout, -1).numpy())
    z = torch.empty((N,) + tuple(z_ref.shape[1:]), dtype=dtype, device=dev)
    y = ops.conv2d_h16(x.to(dev), wf, (k, k, Cin, Cout), b.to(dev), None if scale is None else scale.to(dev),
                       None if shift is None else shift.to(dev), 1, padding, act, z_out=z)
    torch.cuda.synchronize()
    assert y.dtype == dtype and tuple(y.shape) == tuple(y_ref.shape)
    for got, ref, name in ((y, y_ref, "out"), (z, z_ref, "z")):
        err = float((got.float().cpu() - ref).abs().max()) / float(ref.abs().max())
        assert err <= TOL[dtype], "%s: max error %.3g of max |ref| (allowed %.3g)" % (name, err, TOL[dtype])
    # data gradient: the same kernel on dz with the rotated weight image
    if (padding == "same" or k == 1) and Cin % 128 == 0 and not (h16_tile in ("big", "phase", "slab") and Cin % 256):
        xg = x.float().clone().requires_grad_(True)
        yy = orc.conv2d_nhwc(xg, w.float(), None, 1, padding)
        dz = torch.tensor(rng.standard_normal(tuple(yy.shape)).astype(np.float32)).to(dtype)
        yy.backward(dz.float())
        pad = ((k - 1) // 2, (k - 1) // 2) if k > 1 else "valid"
        dx = ops.conv2d_h16(dz.to(dev), wd, (k, k, Cout, Cin), None, None, None, 1, pad, 0)
        torch.cuda.synchronize()
        err = float((dx.float().cpu() - xg.grad).abs().max()) / float(xg.grad.abs().max())
        assert err <= TOL[dtype], "dgrad: max error %.3g (allowed %.3g)" % (err, TOL[dtype])


SLAB_CASES = [
    # N, H, W, Cin, Cout, act, bn, grid (MRCNN_H16P_GRID: workgroups; few of them make every workgroup walk several tiles)
    (400, 14, 14, 256, 256, 1, True, 0),        # 306.25 tiles on 256 workgroups: a second, partial round (its remainder goes to the small-tile kernel)
    (64, 14, 14, 256, 256, 1, True, 5),         # 49 tiles on 5 workgroups: ten tile borders per workgroup, slab of the next tile staged across them
    (37, 14, 14, 128, 512, 0, False, 3),        # two chunks (the minimum), two column tiles per row tile, ragged M (7252), no BN / activation
    (50, 9, 13, 512, 256, 1, True, 4),          # 8 chunks, map 9 x 13 (halo 14 rows), ragged
    (3, 5, 3, 128, 256, 1, False, 0),           # one partial tile (45 rows): every row near a border, tile tail masked
    (30, 2, 14, 256, 256, 0, True, 2),          # maps of two rows: the taps above and below leave the map for every pixel
]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("case", SLAB_CASES)
def test_conv_fwd_h16_slab(dev, case, dtype):
    """conv_fwd_h16q_kernel (the phased 256 x 256 tile with slab-plus-halo staging: a chunk's pixels once, nine taps as shifted
    LDS reads, border taps redirected to a zero row) against the float32 oracle on the same 16-bit-rounded operands: forward
    with the pre-BN output, and the data gradient where its shape is eligible too."""
    ops = _ops()
    N, H, W, Cin, Cout, act, bn, grid = case
    rng = np.random.default_rng(N * 1000 + H * 100 + W * 10 + Cin + Cout)
    x = torch.tensor(rng.standard_normal((N, H, W, Cin)).astype(np.float32)).to(dtype)
    w = torch.tensor((rng.standard_normal((3, 3, Cin, Cout)) / np.sqrt(9 * Cin)).astype(np.float32)).to(dtype)
    b = torch.tensor(rng.standard_normal(Cout).astype(np.float32) * 0.1)
    z_ref = orc.conv2d_nhwc(x.float(), w.float(), b, 1, "same")
    y_ref = z_ref
    scale = shift = None
    if bn:
        scale = torch.tensor(rng.uniform(0.5, 1.5, Cout).astype(np.float32))
        shift = torch.tensor(rng.uniform(-0.2, 0.2, Cout).astype(np.float32))
        y_ref = z_ref * scale + shift
    if act == 1:
        y_ref = torch.relu(y_ref)
    os.environ["MRCNN_H16_TILE"] = "phase"
    if grid:
        os.environ["MRCNN_H16P_GRID"] = str(grid)
    ops.tuning_set("h16_slab", 1)
    try:
        wf, wd = ops.weights_to_h16(w.float().to(dev), dtype)
        z = torch.full((N, H, W, Cout), float("nan"), dtype=dtype, device=dev)
        y = torch.full((N, H, W, Cout), float("nan"), dtype=dtype, device=dev)
        ops.conv2d_h16(x.to(dev), wf, (3, 3, Cin, Cout), b.to(dev), None if scale is None else scale.to(dev),
                       None if shift is None else shift.to(dev), 1, "same", act, z_out=z, out=y)
        torch.cuda.synchronize()
        for got, ref, name in ((y, y_ref, "out"), (z, z_ref, "z")):
            err = float((got.float().cpu() - ref).abs().max()) / float(ref.abs().max())
            assert err <= TOL[dtype], "%s: max error %.3g of max |ref| (allowed %.3g)" % (name, err, TOL[dtype])
        if Cin % 256 == 0 and Cout % 128 == 0:
            xg = x.float().clone().requires_grad_(True)
            yy = orc.conv2d_nhwc(xg, w.float(), None, 1, "same")
            dz = torch.tensor(rng.standard_normal(tuple(yy.shape)).astype(np.float32)).to(dtype)
            yy.backward(dz.float())
            dx = ops.conv2d_h16(dz.to(dev), wd, (3, 3, Cout, Cin), None, None, None, 1, (1, 1), 0)
            torch.cuda.synchronize()
            err = float((dx.float().cpu() - xg.grad).abs().max()) / float(xg.grad.abs().max())
            assert err <= TOL[dtype], "dgrad: max error %.3g (allowed %.3g)" % (err, TOL[dtype])
    finally:
        ops.tuning_set("h16_slab", -1)
        del os.environ["MRCNN_H16_TILE"]
        os.environ.pop("MRCNN_H16P_GRID", None)


def test_float16_overflow_skips_the_step_and_lowers_the_loss_scale(dev):
    """Float16 mode end to end: with an absurd loss scale the 16-bit gradients overflow, the flat gradient buffer's norm is
    not finite, the guarded optimiser step leaves weights and momentum untouched and counts the step on the device;
    adapt_loss_scale (what MaskRCNN.train calls once per epoch) halves the scale per skipped step.  With the scale back in
    range the same batch trains: weights move, nothing is skipped."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_engine_gpu as T
    from caesar_mrcnn_amd.model import MaskRCNN
    cfg = T._small_cfg("resnet50", 128)
    w = T._weights(cfg, 31, damp=0.5)
    inputs, keys = T._train_inputs(cfg, 2, 33)
    model = MaskRCNN("training", cfg, "/tmp/mrcnn_logs", device=dev, weights=w)
    model.compile(0.001, 0.9)
    eng = model.engine
    eng.head_dtype = torch.float16
    eng.loss_scale = 2.0 ** 40                                   # every scaled float16 gradient overflows
    before = eng.params.clone()
    for _ in range(2):
        model.train_on_batch(inputs, rand_keys=keys)
    torch.cuda.synchronize()
    assert torch.equal(eng.params, before), "a step with non-finite gradients changed the weights"
    assert bool(torch.isfinite(eng.params).all()) and float(eng.momentum.abs().max()) == 0.0
    assert eng.skipped_step_count() == 2
    assert eng.adapt_loss_scale() == 2 and eng.loss_scale == 2.0 ** 38
    assert eng.adapt_loss_scale() == 0 and eng.loss_scale == 2.0 ** 38
    eng.loss_scale = 4096.0
    losses = model.train_on_batch(inputs, rand_keys=keys)
    torch.cuda.synchronize()
    assert np.isfinite(losses.cpu().numpy()).all() and eng.skipped_step_count() == 2
    assert not torch.equal(eng.params, before) and bool(torch.isfinite(eng.params).all())


def test_phased_kernels_at_full_size_agree_with_the_reference_kernels(dev):
    """At BASELINE's full mask-head size (2048 ROIs: M = 401 408) and on the one-tile-per-workgroup shape the CPU oracle is too
    slow, so the phased kernels are checked against the independently written 256 x 128 / table-driven kernels (themselves
    checked against the oracle on small shapes above): same exact 16-bit products, float32 accumulation in another order, one
    output rounding -- and by linearity in the input, conv(2x) == 2 conv(x) exactly in binary floating point (up to float16's subnormal step)."""
    ops = _ops()
    torch.manual_seed(5)
    for (N, H, W, Cin, Cout) in ((2048, 14, 14, 256, 256), (4, 128, 128, 256, 256)):
        for dtype in (torch.float16, torch.bfloat16):
            x = torch.randn(N, H, W, Cin, device=dev).to(dtype)
            w = torch.randn(3, 3, Cin, Cout, device=dev) / (3 * Cin ** 0.5)
            wf, _ = ops.weights_to_h16(w, dtype)
            b = torch.randn(Cout, device=dev) * 0.1
            res = {}
            for tile in ("small", "phase"):
                os.environ["MRCNN_H16_TILE"] = tile
                try:
                    z = torch.empty((N, H, W, Cout), dtype=dtype, device=dev)
                    y = ops.conv2d_h16(x, wf, (3, 3, Cin, Cout), b, None, None, 1, "same", 1, z_out=z)
                    y2 = ops.conv2d_h16(x * 2, wf, (3, 3, Cin, Cout), None, None, None, 1, "same", 0)
                    y1 = ops.conv2d_h16(x, wf, (3, 3, Cin, Cout), None, None, None, 1, "same", 0)
                finally:
                    del os.environ["MRCNN_H16_TILE"]
                torch.cuda.synchronize()
                assert bool(torch.isfinite(y.float()).all()) and bool(torch.isfinite(z.float()).all())
                # exact, except where a float16 result is subnormal (|y| < 2^-14: the rounding step there is absolute, 2^-24)
                lin = float((y2.float() - 2 * y1.float()).abs().max())
                assert lin <= (2.0 ** -23 if dtype == torch.float16 else 0.0), "linearity (%s, %s): %.3g" % (tile, dtype, lin)
                res[tile] = (y.float(), z.float())
            for a, r, name in ((res["phase"][0], res["small"][0], "out"), (res["phase"][1], res["small"][1], "z")):
                err = float((a - r).abs().max()) / float(r.abs().max())
                assert err <= TOL[dtype], "%s %s: %.3g" % (name, dtype, err)
            if N == 2048:                       # weight gradient: phased (default at this size) against the table-driven kernel
                dy = torch.randn(N, H, W, Cout, device=dev).to(dtype)
                got = {}
                for mode in ("0", "1"):
                    os.environ["MRCNN_WGRAD_H16_PHASE"] = mode
                    try:
                        got[mode] = ops.conv2d_wgrad_h16(x, dy, (3, 3, Cin, Cout), 1, "same")
                    finally:
                        del os.environ["MRCNN_WGRAD_H16_PHASE"]
                torch.cuda.synchronize()
                err = float((got["1"] - got["0"]).abs().max()) / float(got["0"].abs().max())
                assert err <= 2e-4, "wgrad %s: %.3g" % (dtype, err)


_FIRST_CALL_SCRIPT = r"""
import os, sys
sys.path.insert(0, %(root)r)
import torch
from caesar_mrcnn_amd import ops
dev = torch.device("cuda", 0)
torch.manual_seed(0)
N, H, W, Cin, Cout = 4, 128, 128, 256, 256
x = (torch.randn(N, H, W, Cin, device=dev) * 3).half()
w = torch.randn(3, 3, Cin, Cout, device=dev) / (3 * Cin ** 0.5)
wf, _ = ops.weights_to_h16(w, torch.float16)
b = torch.randn(Cout, device=dev) * 0.1
os.environ["MRCNN_H16_TILE"] = "phase"
y = ops.conv2d_h16(x, wf, (3, 3, Cin, Cout), b, None, None, 1, "same", 1)       # the process's FIRST launch of the kernel
torch.cuda.synchronize()
os.environ["MRCNN_H16_TILE"] = "small"
ref = ops.conv2d_h16(x, wf, (3, 3, Cin, Cout), b, None, None, 1, "same", 1)
torch.cuda.synchronize()
bad = int(((y.float() - ref.float()).abs() > 0.05).sum()) + int((~torch.isfinite(y.float())).sum())
print("first-call-bad", bad)
"""


def test_phased_kernel_first_call_in_a_fresh_process(dev):
    """The store-data hazard of DESIGN.md 5d only showed in the FIRST call of a fresh process (cold instruction cache): about
    one process in ten stored 16-32 stray dwords on exactly this shape.  Three fresh processes, zero wrong elements each (the
    machine-code check in tests/test_host_cpu.py is the deterministic guard; this is the behavioural one)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for _ in range(3):
        r = subprocess.run([sys.executable, "-c", _FIRST_CALL_SCRIPT % {"root": root}], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "first-call-bad 0" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


WGRAD_H16_CASES = [
    # N, H, W, Cin, Cout, k, padding
    (64, 14, 14, 256, 256, 3, "same"),       # mask-head conv, several pixel splits; phased: 7 K-steps per split (odd: one zero step)
    (3, 13, 11, 256, 128, 3, "same"),        # M = 429: pixel tail inside a 32-row step, odd H/W
    (37, 7, 7, 256, 1024, 7, "valid"),       # class-head FC as 7x7 VALID conv: 49 taps, one pixel per ROI
    (5, 16, 16, 512, 128, 1, "valid"),       # 1x1, two channel tiles in one tap
    (3, 13, 11, 256, 256, 3, "same"),        # phased kernel: M = 429 -> one K-step per split, ragged last split, odd H/W
    (5, 16, 16, 512, 256, 1, "valid"),       # phased kernel: 1x1, two input-channel tiles
    (4, 16, 16, 256, 512, 3, "same"),        # phased kernel: two output-channel tiles
    (2, 12, 12, 256, 256, 5, "same"),        # phased kernel: 25 taps, padding 2
]


@pytest.fixture(params=["table", "phase"])
def wgrad_h16_kernel(request):
    """MRCNN_WGRAD_H16_PHASE is read per call: 0 = the table-driven 256 x 128 kernel, 2 = the phased 256 x 256 kernel on every
    shape it can take (stride 1, output as large as the input, channel counts multiples of 256; other shapes fall through)."""
    os.environ["MRCNN_WGRAD_H16_PHASE"] = "0" if request.param == "table" else "2"
    yield request.param
    del os.environ["MRCNN_WGRAD_H16_PHASE"]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("case", WGRAD_H16_CASES)
def test_conv_wgrad_h16(dev, case, dtype, wgrad_h16_kernel):
    """float32 result of exact 16-bit products: only the summation order differs from the oracle (2e-4 of max)."""
    ops = _ops()
    N, H, W, Cin, Cout, k, padding = case
    if wgrad_h16_kernel == "phase" and (Cout % 256 or padding == "valid" and k > 1):
        pytest.skip("phased kernel: Cout % 256 == 0, output as large as the input")
    rng = np.random.default_rng(sum(case[:6]) + 1)
    x = torch.tensor(rng.standard_normal((N, H, W, Cin)).astype(np.float32)).to(dtype)
    w = torch.zeros((k, k, Cin, Cout), requires_grad=True)
    y = orc.conv2d_nhwc(x.float(), w, None, 1, padding)
    dy = torch.tensor(rng.standard_normal(tuple(y.shape)).astype(np.float32)).to(dtype)
    y.backward(dy.float())
    dw = ops.conv2d_wgrad_h16(x.to(dev), dy.to(dev), (k, k, Cin, Cout), 1, padding)
    torch.cuda.synchronize()
    ref = w.grad
    err = float((dw.cpu() - ref).abs().max()) / float(ref.abs().max())
    assert err <= 2e-4, "wgrad: max error %.3g of max |ref|" % err
    dw2 = ops.conv2d_wgrad_h16(x.to(dev), dy.to(dev), (k, k, Cin, Cout), 1, padding, dw=dw.clone(), accumulate=True, multiplier=0.5)
    torch.cuda.synchronize()
    err = float((dw2.cpu() - 1.5 * ref).abs().max()) / float(ref.abs().max())
    assert err <= 3e-4


def test_casts(dev):
    ops = _ops()
    x = torch.randn(100003, device=dev) * 3
    for dtype in (torch.float16, torch.bfloat16):
        h = ops.cast_to_h16(x, dtype)
        assert torch.equal(h, x.to(dtype))
        assert torch.equal(ops.cast_from_h16(h, 0.5), h.float() * 0.5)


@pytest.mark.parametrize("backbone,wide", [("custom", False), ("custom", True), ("resnet50", True)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_mixed_precision_training_step(dev, dtype, backbone, wide):
    """engine.head_dtype: the mask head (wide=False: stages 1-2) and, with engine.h16_wide, also the FPN smoothing
    convolutions, the shared RPN convolution and the two class-head FC layers (stage 3) on the 16-bit matrix cores --
    forward, data and weight gradient -- everything else float32.  Against the all-float32 engine on the same batch
    and the SAME proposals (the float32 run's ProposalLayer output is forced into the 16-bit run: RPN scores that
    differ in the 4th digit re-order the NMS and would change the sampled ROIs, i.e. compare two different problems):
    losses to 1e-2 (f16) / 4e-2 (bf16), every gradient tensor to 4x that of its float32 maximum.
    backbone resnet50 adds stage 4: the identity bottleneck blocks of res4 / res5 (16-bit activations from block to
    block, residual adds and ReLU in the convolution's epilogue, 16-bit data gradients with the shortcut's gradient added
    in the last one)."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_engine_gpu as T
    from caesar_mrcnn_amd.model import MaskRCNN
    cfg = T._small_cfg(backbone, 128)
    w = T._weights(cfg, 31, damp=0.5 if backbone != "custom" else None)
    inputs, keys = T._train_inputs(cfg, 2, 33)
    res = {}
    forced = None
    for mode in (None, dtype):
        model = MaskRCNN("training", cfg, "/tmp/mrcnn_logs", device=dev, weights=w)
        eng = model.engine
        eng.head_dtype, eng.h16_wide = mode, wide
        eng.forced_rpn_rois = forced
        losses = model.train_on_batch(inputs, rand_keys=keys, apply=False, keep_outputs=True)
        torch.cuda.synchronize()
        if mode is None:
            forced = eng.last["rpn_rois"].clone()
        else:
            assert eng._h16_layer("rpn_conv_shared") == wide and eng._h16_layer("mrcnn_class_conv1") == wide
            n16 = sum(eng._h16_block(b) for st in eng.stages for b in st)
            if backbone == "resnet50":
                assert n16 == 16, n16                       # every bottleneck block of ResNet-50
            assert (n16 > 0) == wide
        res[mode] = (losses.cpu().numpy(), eng.get_weights(grads=True))
    tol = 1e-2 if dtype == torch.float16 else 4e-2
    np.testing.assert_allclose(res[dtype][0], res[None][0], rtol=tol)
    bad, l2s = [], []
    for name, ref in res[None][1].items():
        d = res[dtype][1][name].astype(np.float64) - ref
        scale = max(float(np.abs(ref).max()), 1e-10)
        err = float(np.abs(d).max()) / scale
        l2 = float(np.linalg.norm(d)) / max(float(np.linalg.norm(ref)), 1e-12)
        l2s.append(l2)
        if n16 == 0:
            if err > 4 * tol:
                bad.append((name, err))
        elif l2 > 10 * tol:
            # 16-bit activations AND gradients from block to block through all bottleneck blocks: every layer rounds twice
            # (11 / 8 significant bits) and an activation on the ReLU boundary may flip, which moves single elements by
            # 5-15 % of the tensor's maximum (tools/h16_grad_diag.py, ResNet-50: median L2 error 1.8 % f16 / 4.4 % bf16, the
            # largest tensor 3.6 % / 9.8 %; 8.6 % on the 32-pixel res5 maps of the small test backbone; with the blocks in
            # float32 1.2 % / 2.6 %).  The bound is on the L2 error per tensor.
            bad.append((name, l2))
    assert not bad, bad[:6]
    assert np.median(l2s) <= 2.5 * tol, np.median(l2s)
    names = ["mrcnn_mask_conv2/kernel", "rpn_conv_shared/kernel", "fpn_p3/kernel", "mrcnn_class_conv1/kernel"]
    if backbone == "resnet50":
        names += ["res4c_branch2b/kernel", "res5b_branch2a/kernel", "bn4d_branch2c/gamma"]
    for name in names:
        g = res[dtype][1][name]
        assert np.abs(g).max() > 0 and np.isfinite(g).all(), name


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_deconv_and_mask_output_stage_h16(dev, dtype, h16_tile):
    """16-bit deconvolution (pixel-shuffle store; 256 x 128 tile kernel and the phased kernel's row-store form) + mask 1x1
    conv + sigmoid, and the one-pass backward of that stage, against the float32 ops of the package (themselves checked
    against the oracle) on the same 16-bit-rounded tensors."""
    ops = _ops()
    if h16_tile == "big":
        pytest.skip("the ring kernel has no pixel-shuffle store")
    from caesar_mrcnn_amd.params import deconv_keras_to_gemm
    rng = np.random.default_rng(17)
    M, Cd, C = 37, 256, 4
    x = torch.tensor(rng.standard_normal((M, 14, 14, 256)).astype(np.float32)).to(dtype)
    k = (rng.standard_normal((2, 2, Cd, 256)) * 0.05).astype(np.float32)                  # Keras (2,2,out,in)
    wg = torch.tensor(deconv_keras_to_gemm(k).reshape(256, 4 * Cd)).to(dtype)            # GEMM matrix [Cin, 4*Cd], rounded
    b = torch.tensor(rng.standard_normal(Cd).astype(np.float32) * 0.1)
    up_ref = ops.deconv2x2(x.float().to(dev), wg.float().to(dev), b.to(dev), 1)           # float32 kernel, same operands
    wt, _ = ops.weights_to_h16(wg.float().to(dev).view(1, 1, 256, 4 * Cd), dtype, want_dgrad=False)
    up = ops.deconv2x2_h16(x.to(dev), wt, b.to(dev), Cd, 1)
    torch.cuda.synchronize()
    assert up.dtype == dtype and tuple(up.shape) == (M, 28, 28, Cd)
    err = float((up.float() - up_ref).abs().max()) / float(up_ref.abs().max())
    assert err <= TOL[dtype], err
    wm = torch.tensor(rng.standard_normal((Cd, C)).astype(np.float32) * 0.1, device=dev)
    bm = torch.tensor(rng.standard_normal(C).astype(np.float32) * 0.1, device=dev)
    m_ref = ops.conv2d(up.float(), wm.view(1, 1, Cd, C), bm, stride=1, padding="valid", act=2)
    m = ops.mask_out_fwd_h16(up, wm, bm)
    torch.cuda.synchronize()
    torch.testing.assert_close(m, m_ref, rtol=1e-5, atol=1e-5)
    # backward of the output stage
    g = torch.tensor(rng.standard_normal((M, 28, 28, C)).astype(np.float32) * 1e-3, device=dev)
    ref_dw, ref_dbm, ref_dbd = torch.zeros_like(wm), torch.zeros(C, device=dev), torch.zeros(Cd, device=dev)
    dzg_ref = ops.mask_out_bwd(g, m, up.float(), wm, ref_dw, ref_dbm, ref_dbd)
    dw, dbm, dbd = torch.zeros_like(wm), torch.zeros(C, device=dev), torch.zeros(Cd, device=dev)
    S = 1024.0
    dzg = ops.mask_out_bwd_h16(g, m, up, wm, dw, dbm, dbd, S)
    torch.cuda.synchronize()
    assert dzg.dtype == dtype
    err = float((dzg.float() / S - dzg_ref).abs().max()) / float(dzg_ref.abs().max())
    assert err <= TOL[dtype], err
    for a, r in ((dw, ref_dw), (dbm, ref_dbm), (dbd, ref_dbd)):
        torch.testing.assert_close(a, r, rtol=1e-4, atol=1e-4 * float(r.abs().max()))


def test_graph_replay_follows_weight_updates_h16(dev):
    """HIP-graph inference with the 16-bit mask head: a replay after set_weights (or a training step) must use the new
    weights.  The 16-bit weight images are refreshed in place outside the graph, so their captured addresses stay
    valid; the precision switch is part of the graph key (replaying a float32 capture with head_dtype set, or the
    reverse, would silently run the wrong precision)."""
    import test_engine_gpu as T
    from caesar_mrcnn_amd.model import MaskRCNN
    cfg = T._small_cfg("custom", 128, mode="inference")
    w1, w2 = T._weights(cfg, 3), T._weights(cfg, 4)
    model = MaskRCNN("inference", cfg, "/tmp/mrcnn_logs", device=dev, weights=w1)
    eng = model.engine
    eng.head_dtype = torch.float16
    rng = np.random.default_rng(2)
    x = torch.tensor(rng.uniform(0, 255, (1, 128, 128, 3)).astype(np.float32), device=dev)
    win = torch.tensor([[0., 0., 1., 1.]], device=dev)
    keys = ("detections", "mrcnn_mask")
    e1 = {k: eng.infer(x, win)[k].cpu().numpy().copy() for k in keys}
    g1 = {k: v.cpu().numpy().copy() for k, v in eng.infer_graphed(x, win).items() if k in keys}
    for k in keys:
        assert np.array_equal(e1[k], g1[k]), k
    eng.set_weights(w2)                                      # invalidates the 16-bit images
    e2 = {k: eng.infer(x, win)[k].cpu().numpy().copy() for k in keys}
    eng.set_weights(w2)                                      # again, so that the replay itself has to refresh them
    g2 = {k: v.cpu().numpy().copy() for k, v in eng.infer_graphed(x, win).items() if k in keys}
    assert not np.array_equal(e1["mrcnn_mask"], e2["mrcnn_mask"])
    for k in keys:
        assert np.array_equal(e2[k], g2[k]), k
    eng.head_dtype = None                                    # float32 again: must not replay the 16-bit capture
    e3 = eng.infer(x, win)["mrcnn_mask"].cpu().numpy().copy()
    g3 = eng.infer_graphed(x, win)["mrcnn_mask"].cpu().numpy()
    assert np.array_equal(e3, g3) and not np.array_equal(e3, e2["mrcnn_mask"])


SMALL_CASES = [
    # N, H, W, Cin, Cout, k, stride, padding, act, bn, res
    (4, 32, 32, 1024, 256, 1, 1, "valid", 1, True, False),     # res4 2a at 512 x 512: 16 K-steps of 64
    (4, 32, 32, 256, 256, 3, 1, "same", 1, True, False),       # res4 2b: 36 K-steps, padded taps
    (4, 32, 32, 256, 1024, 1, 1, "valid", 1, True, True),      # res4 2c: shortcut added before the ReLU
    (2, 16, 16, 64, 64, 3, 1, "same", 1, True, False),         # res2-sized: Cin = Cout = 64, one chunk per tap
    (2, 16, 16, 512, 256, 1, 2, "valid", 1, True, False),      # stride-2 1x1 (first block of a stage)
    (3, 13, 11, 128, 192, 3, 1, "same", 0, False, True),       # ragged M (429 = 6 x 64 + 45), 3 column tiles, residual, no BN
    (1, 5, 5, 64, 64, 1, 1, "valid", 2, False, False),         # one K-step (fewer than ring slots), sigmoid
]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("case", SMALL_CASES)
def test_conv_fwd_h16_small_tile(dev, case, dtype):
    """conv_fwd_h16s_kernel (64 x 64 tiles, 4-stage LDS-DMA ring; the trunk's layers) against the float32 oracle on the
    same 16-bit-rounded operands: forward with bias / frozen BN / 16-bit residual / activation, pre-BN output, and the
    data gradient of a stride-2 1x1 convolution as a strided (scattering) store."""
    ops = _ops()
    N, H, W, Cin, Cout, k, stride, padding, act, bn, use_res = case
    rng = np.random.default_rng(sum(v if isinstance(v, int) else 3 for v in case))
    x = torch.tensor(rng.standard_normal((N, H, W, Cin)).astype(np.float32)).to(dtype)
    w = torch.tensor((rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)).to(dtype)
    b = torch.tensor(rng.standard_normal(Cout).astype(np.float32) * 0.1)
    z_ref = orc.conv2d_nhwc(x.float(), w.float(), b, stride, padding)
    y_ref = z_ref
    scale = shift = None
    if bn:
        scale = torch.tensor(rng.uniform(0.5, 1.5, Cout).astype(np.float32))
        shift = torch.tensor(rng.uniform(-0.2, 0.2, Cout).astype(np.float32))
        y_ref = z_ref * scale + shift
    res = None
    if use_res:
        res = torch.tensor(rng.standard_normal(tuple(z_ref.shape)).astype(np.float32)).to(dtype)
        y_ref = y_ref + res.float()
    if act == 1:
        y_ref = torch.relu(y_ref)
    elif act == 2:
        y_ref = torch.sigmoid(y_ref)
    os.environ["MRCNN_H16_SMALL"] = "1"
    try:
        assert ops.conv2d_h16_supported(tuple(x.shape), (k, k, Cin, Cout), stride, padding, res=use_res)
        wf, wd = ops.weights_to_h16(w.float().to(dev), dtype)
        z = torch.empty(tuple(z_ref.shape), dtype=dtype, device=dev)
        y = ops.conv2d_h16(x.to(dev), wf, (k, k, Cin, Cout), b.to(dev), None if scale is None else scale.to(dev),
                           None if shift is None else shift.to(dev), stride, padding, act, z_out=z,
                           res=None if res is None else res.to(dev))
        torch.cuda.synchronize()
        for got, ref, name in ((y, y_ref, "out"), (z, z_ref, "z")):
            err = float((got.float().cpu() - ref).abs().max()) / float(ref.abs().max())
            assert err <= TOL[dtype], "%s: max error %.3g of max |ref| (allowed %.3g)" % (name, err, TOL[dtype])
        if k == 1 and stride == 2:
            # data gradient of the stride-2 1x1 convolution: the 1x1 product scattered to the even pixels of a zeroed tensor
            dz = torch.tensor(rng.standard_normal(tuple(z_ref.shape)).astype(np.float32)).to(dtype)
            xg = x.float().clone().requires_grad_(True)
            orc.conv2d_nhwc(xg, w.float(), None, stride, padding).backward(dz.float())
            dx = torch.zeros((N, H, W, Cin), dtype=dtype, device=dev)
            ops.conv2d_h16(dz.to(dev), wd, (1, 1, Cout, Cin), None, None, None, 1, "valid", 0, out=dx,
                           out_strides=(H * W * Cin, 2 * W * Cin, 2 * Cin))
            torch.cuda.synchronize()
            err = float((dx.float().cpu() - xg.grad).abs().max()) / float(xg.grad.abs().max())
            assert err <= TOL[dtype], "strided dgrad: %.3g" % err
    finally:
        del os.environ["MRCNN_H16_SMALL"]


@pytest.mark.parametrize("dtype,loss_rtol,l2_max,l2_median", [(torch.float16, 5e-3, 8e-2, 2.5e-2), (torch.bfloat16, 2e-2, 2.5e-1, 8e-2)],
                         ids=["float16", "bfloat16"])
def test_cfg5_r101_512_f16_training_step_vs_oracle(dev, dtype, loss_rtol, l2_max, l2_median):
    """BASELINE configs[4] at its real sizes: ResNet-101+FPN 512x512, nimg_per_gpu = 4, 512 train ROIs, 2000 proposals (round 3:
    all four images of a rank's batch and both 16-bit types; round 2 ran two images in float16), engine in its widest 16-bit
    mode (mask head, FPN smoothing, shared RPN convolution, class-head FC layers and all 33 bottleneck blocks on the 16-bit
    MFMA; float32 master weights / accumulation / gradients, loss scale 4096 for float16).  Against the FLOAT32 oracle's
    autograd on the ROIs and targets the engine sampled (fed to the oracle, as in test_cfg2 / cfg3):
      float16:  losses rtol 5e-3; every parameter gradient L2 error <= 8e-2 of its norm, median over tensors <= 2.5e-2
                (measured: 2.1e-4; max 5.6e-2; median 1.2e-2)
      bfloat16: 2e-2; 2.5e-1; 8e-2 (8 significant bits per stored activation / gradient instead of 11; measured 2.2e-3; 1.25e-1; 4.8e-2)
    (ReLU-boundary flips move single elements by more, which is why the bound is on the L2 norm -- see
    test_mixed_precision_training_step)."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_engine_gpu as T
    from caesar_mrcnn_amd.model import MaskRCNN
    cfg = T._full_cfg("resnet101", 512, nimg=4)
    B = 4
    w = T._weights(cfg, 61, damp=0.25)
    inputs, keys = T._train_inputs(cfg, B, 63)
    images, meta, rpn_match, rpn_bbox_t, gt_cls, gt_boxes, gt_masks = inputs
    model = MaskRCNN("training", cfg, "/tmp/mrcnn_logs", device=dev, weights=w)
    eng = model.engine
    eng.head_dtype = dtype
    eng.sparse_mask_bwd = True
    losses = model.train_on_batch(inputs, rand_keys=keys, apply=False, keep_outputs=True)
    torch.cuda.synchronize()
    assert sum(eng._h16_block(b) for st in eng.stages for b in st) == 33          # every bottleneck block of ResNet-101
    last = {k: v.cpu().numpy() for k, v in eng.last.items() if torch.is_tensor(v)}
    # positives in (nearly) every image: with random weights one of four tiles may get no proposal of IoU >= 0.5 -- it then
    # contributes no ROI at all (DetectionTargetLayer's ratio rule), which the oracle reproduces from the forced targets
    assert (last["counts"][:, 0] > 0).sum() >= B - 1 and last["counts"][:, 0].sum() >= 8, last["counts"]
    eng.apply_gradients(0.0, 0.0, world_size=1)
    torch.cuda.synchronize()
    g = eng.get_weights(grads=True)
    names = list(eng.layout.offsets)
    del model, eng
    torch.cuda.empty_cache()
    o = orc.OracleMaskRCNN(cfg, w, requires_grad=True)
    forced = {k: last[k] for k in ("rois", "target_class_ids", "target_bbox", "target_mask")}
    ref = o.forward_training(images, rpn_match, rpn_bbox_t.astype(np.float32), gt_cls, gt_boxes, gt_masks,
                             meta[:, 12:].astype(np.int32), orc.get_anchors(cfg, images.shape[1:]), keys, forced=forced)
    o.total_loss(ref["losses"]).backward()
    want = np.array([float(l.detach()) for l in ref["losses"]])
    l2s, bad = [], []
    for name in names:
        rg = o.w[name].grad.numpy().astype(np.float64)
        d = g[name].astype(np.float64) - rg
        l2 = float(np.linalg.norm(d)) / max(float(np.linalg.norm(rg)), 1e-12)
        l2s.append(l2)
        if l2 > l2_max:
            bad.append((name, l2))
    print("[cfg5 %s, %d images] losses max rel error %.2e; gradient L2 error over %d tensors: median %.2e, max %.2e" % (
        dtype, B, float(np.max(np.abs(losses.cpu().numpy() - want) / np.maximum(np.abs(want), 1e-4))), len(l2s), np.median(l2s), np.max(l2s)))
    np.testing.assert_allclose(losses.cpu().numpy(), want, rtol=loss_rtol, atol=1e-4)
    assert not bad, bad[:8]
    assert np.median(l2s) <= l2_median, np.median(l2s)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("k,cin,cout,use_res", [(1, 1024, 256, False), (3, 256, 256, False), (1, 256, 1024, True), (3, 64, 64, False)])
def test_conv_dgrad_ep_h16_equals_two_launches(dev, dtype, k, cin, cout, use_res):
    """mrcnn_conv2d_dgrad_ep_h16 == mrcnn_conv2d_fwd_h16 (small tile) followed by mrcnn_epilogue_bwd_h16: the fused store rounds
    y * act' * scale once where the pair rounds y first, so dz agrees to one 16-bit rounding (2^-10 / 2^-7 of the maximum) and
    the float32 channel sums to the same; dy (= y * act') likewise."""
    ops = _ops()
    rng = np.random.default_rng(k * 1000 + cin + cout)
    N, H, W = 3, 13, 11                                         # M = 429: ragged against the 64-row tile
    S = 1024.0
    dz = torch.tensor(rng.standard_normal((N, H, W, cin)).astype(np.float32), device=dev).to(dtype)
    w = torch.tensor((rng.standard_normal((k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32), device=dev)
    wf, _ = ops.weights_to_h16(w, dtype)
    y_below = torch.relu(torch.tensor(rng.standard_normal((N, H, W, cout)).astype(np.float32), device=dev)).to(dtype)
    z_below = torch.tensor(rng.standard_normal((N, H, W, cout)).astype(np.float32), device=dev).to(dtype)
    scale, mean, rstd = (torch.tensor(rng.uniform(0.5, 1.5, cout).astype(np.float32), device=dev) for _ in range(3))
    res = torch.tensor(rng.standard_normal((N, H, W, cout)).astype(np.float32), device=dev).to(dtype) if use_res else None
    pad = ((k - 1) // 2, (k - 1) // 2) if k > 1 else "valid"
    os.environ["MRCNN_H16_SMALL"] = "1"
    try:
        sums = [torch.zeros(cout, device=dev) for _ in range(3)]
        got = ops.conv2d_dgrad_ep_h16(dz, wf, (k, k, cin, cout), pad, y_below, z_below, scale, mean, rstd, sums[0], sums[1], sums[2], 1,
                                      1.0 / S, res=res, want_dy=True)
        assert got is not None
        y = ops.conv2d_h16(dz, wf, (k, k, cin, cout), None, None, None, 1, pad, 0, res=res)
        rs = [torch.zeros(cout, device=dev) for _ in range(3)]
        ref, ref_dy = ops.epilogue_bwd_h16(y, y_below, z_below, scale, mean, rstd, rs[0], rs[1], rs[2], 1, 1.0 / S, want_dy=True)
        torch.cuda.synchronize()
    finally:
        del os.environ["MRCNN_H16_SMALL"]
    for a, b, name in ((got[0], ref, "dz"), (got[1], ref_dy, "dy")):
        err = float((a.float() - b.float()).abs().max()) / float(b.float().abs().max())
        assert err <= 2 * TOL[dtype], (name, err)
    for a, b in zip(sums, rs):
        torch.testing.assert_close(a, b, rtol=4 * TOL[dtype], atol=4 * TOL[dtype] * float(b.abs().max()))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("pool", [7, 14])
def test_roialign_h16_fwd_bwd(dev, dtype, pool):
    """mrcnn_roialign_fwd_h16 / _bwd_h16 (configs[4]: the ROI heads gather from the 16-bit pyramid) against the float32 ORACLE
    (PyramidROIAlign, mrcnn/model.py:428-534) evaluated on the same 16-bit-rounded inputs: the interpolation runs in float32 and
    is rounded once, so forward outputs agree to one rounding of the result (2^-10 / 2^-7 relative to the largest corner
    value); the adjoint adds float32 atomics of (16-bit gradient x multiplier) -- equal to the oracle's gradient of the rounded
    dout up to float32 summation order.  ROIs on all four levels, one partly outside the map, zero-padded rows."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_kernels_gpu as K
    ops = _ops()
    rng = np.random.default_rng(pool + (0 if dtype == torch.float16 else 100))
    B, R, C = 2, 60, 256
    area = 1024.0 * 1024.0
    fm16 = [torch.tensor(K._rand(rng, B, s, s, C)).to(dtype) for s in (64, 32, 16, 8)]
    fms = [f.float().requires_grad_(True) for f in fm16]                       # the oracle sees exactly the rounded values
    rois = K._random_rois(rng, B, R, zero_tail=3)
    rois[0, 0] = [0.2, 0.2, 1.2, 0.7]
    ref = orc.pyramid_roi_align(rois, fms, pool, area)
    got = ops.roialign_h16(torch.tensor(rois, device=dev), [f.to(dev) for f in fm16], pool, area)
    torch.cuda.synchronize()
    assert got.dtype == dtype and tuple(got.shape) == tuple(ref.shape)
    eps = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    err = (got.float().cpu() - ref.detach()).abs().max().item()
    assert err <= eps * float(max(f.abs().max() for f in fms)) * 1.01, err
    S = 64.0
    dout16 = (torch.tensor(K._rand(rng, *ref.shape)) * S).to(dtype)
    dout16[1, 5] = 0                                                              # an all-zero row: skipped by the adjoint
    ref.backward(dout16.float() / S)
    dfm = [torch.full((B, s, s, C), 0.5, device=dev) for s in (64, 32, 16, 8)]   # accumulates onto what the maps hold
    ops.roialign_bwd_h16(torch.tensor(rois, device=dev), dout16.to(dev), dfm, pool, area, multiplier=1.0 / S)
    torch.cuda.synchronize()
    for a, f in zip(dfm, fms):
        torch.testing.assert_close(a.cpu() - 0.5, f.grad, rtol=1e-4, atol=1e-4)
