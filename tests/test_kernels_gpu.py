"""Kernel-level parity: every C-ABI entry point against the CPU oracle on seeded inputs (sizes the
oracle finishes in seconds).  Float tolerances are written next to each check; index outputs are
compared exactly."""
import os
import sys

import numpy as np
import pytest
import torch

import mrcnn_oracle as orc

pytestmark = pytest.mark.gpu

F32 = dict(rtol=2e-4, atol=2e-4)     # fp32 contractions with a different summation order


def _ops():
    from caesar_mrcnn_amd import ops
    return ops


def _rand(rng, *shape, scale=1.0):
    return (rng.standard_normal(shape) * scale).astype(np.float32)


def _cfg(**kw):
    from caesar_mrcnn_amd.config import run_py_config
    return run_py_config(**kw)


# ------------------------------------------------------------------------------------------------
CONV_CASES = [
    # N, H, W, Cin, Cout, k, stride, padding, act, bn, res
    (2, 16, 16, 64, 256, 1, 1, "valid", 1, True, 0),
    (3, 14, 14, 256, 256, 3, 1, "same", 1, True, 0),
    (2, 16, 16, 256, 128, 1, 2, "valid", 1, True, 0),
    (2, 64, 64, 3, 64, 7, 2, (3, 3), 1, True, 0),        # stem: generic gather path, K = 147
    (2, 16, 16, 64, 256, 1, 1, "valid", 1, True, 1),     # residual add + relu
    (2, 16, 16, 512, 256, 1, 1, "valid", 0, False, 2),   # FPN lateral + nearest-up2 add
    (5, 7, 7, 256, 1024, 7, 1, "valid", 1, True, 0),     # class head FC as 7x7 VALID conv, K = 12544
    (1, 33, 17, 64, 64, 3, 1, "same", 0, False, 0),      # ragged M, 64x64 tile
    (8, 64, 64, 64, 256, 1, 1, "valid", 1, True, 0),     # 128x128 tiles
    (8, 64, 64, 64, 64, 3, 1, "same", 1, True, 0),       # 128x64 tiles
    (16, 64, 64, 256, 12, 1, 1, "valid", 0, False, 0),   # 128x32 tiles, ragged Cout
    (1, 8, 8, 512, 6, 1, 1, "valid", 0, False, 0),       # 64x32 tiles, scalar B loads
    (4, 28, 28, 256, 4, 1, 1, "valid", 2, False, 0),     # mask logits + sigmoid
    (64, 14, 14, 256, 256, 3, 1, "same", 1, True, 1),    # LDS-DMA 128x128 kernel: padded taps, residual
    (90, 13, 11, 128, 256, 3, 1, "same", 1, False, 0),   # LDS-DMA kernel, ragged M (12870 = 100 x 128 + 70), odd H/W
    (3, 64, 64, 256, 128, 1, 1, "valid", 0, False, 0),   # LDS-DMA kernel, 1x1, one N tile
    (457, 14, 14, 64, 256, 3, 1, "same", 1, True, 1),    # LDS-DMA kernel with the TAIL SPLIT: 700 x 2 tiles = 1.09 rounds of 1280 slots ->
                                                         # 640 row tiles unsplit, 60 (ragged M = 89572) as 4 K slices + slab reduction; BN, residual, z
    # conv_fwd_sk16_kernel (16 x 16 tiles, four waves split K): the batch-1 detect shapes, too small for the 32 x 32 single-launch kernel
    (1, 16, 16, 1024, 256, 1, 1, "valid", 1, True, 0),   # res4 2a at batch 1: 256 tiles, 8 K-steps per wave
    (1, 16, 16, 256, 256, 3, 1, "same", 1, True, 0),     # res4 2b: padded taps, 18 K-steps per wave
    (1, 16, 16, 256, 1024, 1, 1, "valid", 1, True, 1),   # res4 2c: K = 256 (2 K-steps per wave), shortcut added before the ReLU
    (1, 8, 8, 2048, 512, 1, 1, "valid", 1, True, 0),     # res5 2a: 128 tiles
    (1, 7, 9, 256, 128, 3, 1, "same", 0, False, 1),      # ragged M (63 = 3 x 16 + 15), odd map, residual, no BN
    (1, 16, 16, 512, 256, 1, 2, "valid", 1, True, 0),    # stride-2 1 x 1 (first block of a stage): 64 pixels out
]


@pytest.fixture
def tail_split_env():
    """MRCNN_CONV_TAIL_SPLIT is read per call (default off: measured slower inside the step); on for the shapes that test it."""
    os.environ["MRCNN_CONV_TAIL_SPLIT"] = "1"
    yield
    del os.environ["MRCNN_CONV_TAIL_SPLIT"]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd(dev, case, tail_split_env):
    ops = _ops()
    N, H, W, Cin, Cout, k, stride, padding, act, bn, res = case
    rng = np.random.default_rng(1000 + sum(int(v) if isinstance(v, int) else 7 for v in case))
    x = _rand(rng, N, H, W, Cin)
    w = _rand(rng, k, k, Cin, Cout, scale=1.0 / np.sqrt(k * k * Cin))
    b = _rand(rng, Cout, scale=0.1)
    xt, wt, bt = (torch.tensor(a, device=dev) for a in (x, w, b))
    y = orc.conv2d_nhwc(torch.tensor(x), torch.tensor(w), torch.tensor(b), stride, padding)
    z_ref = y.clone()
    scale = shift = None
    if bn:
        g, be, mu, var = rng.uniform(.5, 1.5, Cout), rng.uniform(-.2, .2, Cout), rng.uniform(-.2, .2, Cout), rng.uniform(.5, 1.5, Cout)
        g, be, mu, var = (torch.tensor(a.astype(np.float32)) for a in (g, be, mu, var))
        y = orc.batchnorm_frozen(y, g, be, mu, var)
        scale = torch.empty(Cout, device=dev); shift = torch.empty(Cout, device=dev); rstd = torch.empty(Cout, device=dev)
        ops.bn_fold(g.to(dev), be.to(dev), mu.to(dev), var.to(dev), scale, shift, rstd)
        torch.testing.assert_close(rstd.cpu(), 1.0 / torch.sqrt(var + orc.BN_EPS), rtol=1e-6, atol=1e-6)
    rt = None
    if res == 1:
        r = _rand(rng, *y.shape)
        y = y + torch.tensor(r)
        rt = torch.tensor(r, device=dev)
    elif res == 2:
        r = _rand(rng, N, y.shape[1] // 2, y.shape[2] // 2, Cout)
        y = y + torch.tensor(r).repeat_interleave(2, 1).repeat_interleave(2, 2)
        rt = torch.tensor(r, device=dev)
    if act == 1:
        y = torch.relu(y)
    elif act == 2:
        y = torch.sigmoid(y)
    z = torch.empty(tuple(y.shape), device=dev)
    ops.tuning_set("sk16", 1)           # what engine.infer sets: the 16 x 16-tile kernel may take the shapes written for it below
    try:
        out = ops.conv2d(xt, wt, bt, scale, shift, rt, stride, padding, act, res, z_out=z)
    finally:
        ops.tuning_set("sk16", 0)
    torch.cuda.synchronize()
    torch.testing.assert_close(out.cpu(), y, **F32)
    torch.testing.assert_close(z.cpu(), z_ref, **F32)


@pytest.mark.parametrize("case", [(2, 37, 14, 14, 64, 128), (2, 5, 8, 12, 32, 256), (2, 130, 14, 14, 256, 256),
                                  (4, 37, 14, 14, 64, 128), (4, 5, 7, 9, 32, 256), (4, 130, 14, 14, 256, 256), (4, 3, 16, 8, 128, 128),
                                  (6, 37, 14, 14, 64, 128), (6, 9, 6, 10, 32, 256), (6, 130, 14, 14, 256, 256)])
def test_conv_winograd(dev, case):
    """Winograd F(2x2, 3x3) / F(4x4, 3x3) path of the 3x3 'same' convolutions (input transform, 16 / 36 batched GEMMs in one launch,
    output transform with bias / frozen BN / ReLU / pre-BN z): against the oracle -- tile 2 at the float32 tolerance of the direct
    kernels (its transforms use +-1 and 1/2 only), tile 4 at 5e-5 of the result's range (constants up to 8: one decimal digit;
    measured 1.3e-5 at K = 256).  Tile counts that are not multiples of 128 exercise the padded GEMM rows; 14, 7 and 9 are not
    multiples of 4 (tiles that hang over the edge).  Tile 6 = TILE_MIXED: extents that are 2 mod 4 covered by 4 x 4 tiles and a
    last row / column of 4 x 2, 2 x 4 and 2 x 2 tiles (four groups, no overhang; same tolerance as tile 4).  Then the weight
    gradient through the same domain and the data-gradient form fused with the epilogue backward of the layer below, against the
    direct kernels of the package."""
    ops = _ops()
    tile, N, H, W, Cin, Cout = case
    rng = np.random.default_rng(300 + sum(case))
    x = _rand(rng, N, H, W, Cin)
    w = _rand(rng, 3, 3, Cin, Cout, scale=1.0 / np.sqrt(9 * Cin))
    b = _rand(rng, Cout, scale=0.1)
    sc = rng.uniform(.5, 1.5, Cout).astype(np.float32); sh = rng.uniform(-.2, .2, Cout).astype(np.float32)
    z_ref = orc.conv2d_nhwc(torch.tensor(x), torch.tensor(w), torch.tensor(b), 1, "same")
    y_ref = torch.relu(z_ref * torch.tensor(sc) + torch.tensor(sh))
    xt, wt, bt, sct, sht = (torch.tensor(a, device=dev) for a in (x, w, b, sc, sh))
    U = ops.winograd_weights(wt, tile=tile)
    if tile == ops.TILE_MIXED:
        assert [tuple(u.shape) for u in U] == [(nb, Cin, Cout) for nb in (36, 24, 24, 16)]
    else:
        assert tuple(U.shape) == ((tile + 2) ** 2, Cin, Cout)
    tol = lambda ref: F32 if tile == 2 else dict(rtol=1e-4, atol=5e-5 * float(ref.abs().max()))
    z = torch.empty((N, H, W, Cout), device=dev)
    y = ops.conv2d_winograd(xt, U, bt, sct, sht, 1, z_out=z)
    torch.cuda.synchronize()
    torch.testing.assert_close(y.cpu(), y_ref, **tol(y_ref))
    torch.testing.assert_close(z.cpu(), z_ref, **tol(z_ref))
    V = torch.empty(ops.winograd_v_floats((N, H, W, Cin), tile), device=dev)
    y0 = ops.conv2d_winograd(xt, U, keep_v=V)                    # no epilogue at all (the last data gradient of the chain)
    torch.cuda.synchronize()
    torch.testing.assert_close(y0.cpu(), z_ref - torch.tensor(b), **tol(z_ref))
    # weight gradient through the same domain: V kept from the forward pass, dz transformed by the output transform's adjoint
    xg = torch.tensor(x); wg = torch.tensor(w, requires_grad=True)
    yy = orc.conv2d_nhwc(xg, wg, None, 1, "same")
    dy = _rand(rng, N, H, W, Cout)
    yy.backward(torch.tensor(dy))
    dw = torch.full((3, 3, Cin, Cout), 7.0, device=dev)
    ops.conv2d_wgrad_winograd(V, (N, H, W, Cin), torch.tensor(dy, device=dev), dw, tile=tile)
    torch.cuda.synchronize()
    assert float((dw.cpu() - wg.grad).abs().max()) <= 5e-4 * float(wg.grad.abs().max())
    ops.conv2d_wgrad_winograd(V, (N, H, W, Cin), torch.tensor(dy, device=dev), dw, accumulate=True, tile=tile)
    torch.cuda.synchronize()
    assert float((dw.cpu() - 2 * wg.grad).abs().max()) <= 1e-3 * float(wg.grad.abs().max())
    if Cin % 128:
        return
    # data gradient of a Cout -> Cin layer + epilogue backward of the layer below (Cin channels, frozen BN, ReLU)
    dz = torch.tensor(_rand(rng, N, H, W, Cout), device=dev)
    wflip = torch.empty((3, 3, Cout, Cin), device=dev)
    ops.weight_flip_transpose(wt, wflip)
    below_out = torch.relu(torch.tensor(_rand(rng, N, H, W, Cin), device=dev))
    below_z = torch.tensor(_rand(rng, N, H, W, Cin), device=dev)
    scale, mean, rstd = (torch.tensor(rng.uniform(0.5, 1.5, Cin).astype(np.float32), device=dev) for _ in range(3))
    sums = [torch.zeros(Cin, device=dev) for _ in range(3)]
    got = ops.conv2d_dgrad_ep_winograd(dz, ops.winograd_weights(wflip, tile=tile), below_out, below_z, scale, mean, rstd, sums[0], sums[1], sums[2], 1)
    yd = ops.conv2d(dz, wflip, stride=1, padding=(1, 1))
    ref = torch.empty_like(yd)
    rs = [torch.zeros(Cin, device=dev) for _ in range(3)]
    ops.epilogue_bwd(yd, below_out, below_z, scale, mean, rstd, None, ref, rs[0], rs[1], rs[2], 1)
    torch.cuda.synchronize()
    torch.testing.assert_close(got, ref, **tol(ref))
    for a, r in zip(sums, rs):
        torch.testing.assert_close(a, r, rtol=2e-4, atol=2e-4 * float(r.abs().max()))
    # round 3: the ReLU mask recomputed from the stored z (mrcnn_winograd_output_bwd_zmask_g) instead of read from the
    # activated output: with (z, out) a real forward pair -- out = max(scale * z + shift, 0), including values exactly on the
    # boundary -- dz must equal the out-reading form bit for bit, and the channel sums up to the order of their atomics
    shift = torch.tensor(rng.uniform(-0.3, 0.3, Cin).astype(np.float32), device=dev)
    zf = torch.tensor(_rand(rng, N, H, W, Cin), device=dev)
    zf[0, 0, 0, :] = -shift / scale                                   # scale * z + shift == 0 (or within an ulp of it)
    outf = torch.clamp_min(scale * zf + shift, 0.0)                   # the forward epilogue's expression (torch: mul, add, max -- no fma)
    Ut = ops.winograd_weights(wflip, tile=tile)
    s1 = [torch.zeros(Cin, device=dev) for _ in range(3)]
    s2 = [torch.zeros(Cin, device=dev) for _ in range(3)]
    a_ = ops.conv2d_dgrad_ep_winograd(dz, Ut, outf, zf, scale, mean, rstd, s1[0], s1[1], s1[2], 1)
    b_ = ops.conv2d_dgrad_ep_winograd(dz, Ut, None, zf, scale, mean, rstd, s2[0], s2[1], s2[2], 1, fwd_shift=shift)
    torch.cuda.synchronize()
    assert torch.equal(a_, b_) and float(a_.abs().max()) > 0 and float((a_ == 0).float().mean()) > 0.2
    for u, v in zip(s1, s2):
        torch.testing.assert_close(u, v, rtol=1e-5, atol=1e-5 * float(v.abs().max()))


@pytest.mark.parametrize("tile", [4, 6])
def test_winograd_fused_output_equals_unfused(dev, monkeypatch, tile):
    """mrcnn_winograd_gemm_fused (round 3): the 4 x 4 group's output transform inside the GEMM launch -- row-block-major tile
    order, per-row-block arrival counters, the last arriver transforms -- against the separate launches on a training-sized layer
    (640 ROIs of 14 x 14 x 256: 5760 GEMM tiles, 160 row blocks of which the last is ragged for the uniform tiling).  The
    transform code and the per-tile arithmetic are the same, so: forward out and z BIT-identical; data gradient dz of the layer
    below bit-identical (ReLU mask from `out` and from z), channel sums equal up to the order of their float atomics.  Run
    three times over: the counters must come back to zero and a stale Mt of the previous call must never be read."""
    ops = _ops()
    rng = np.random.default_rng(5 + tile)
    N, H, W, C_ = 640, 14, 14, 256
    x = torch.tensor(_rand(rng, N, H, W, C_), device=dev)
    w = torch.tensor(_rand(rng, 3, 3, C_, C_, scale=0.03), device=dev)
    b = torch.tensor(_rand(rng, C_, scale=0.1), device=dev)
    sc = torch.tensor(rng.uniform(.5, 1.5, C_).astype(np.float32), device=dev)
    sh = torch.tensor(rng.uniform(-.2, .2, C_).astype(np.float32), device=dev)
    mean, rstd = (torch.tensor(rng.uniform(0.5, 1.5, C_).astype(np.float32), device=dev) for _ in range(2))
    U = ops.winograd_weights(w, tile=tile)
    wflip = torch.empty((3, 3, C_, C_), device=dev)
    ops.weight_flip_transpose(w, wflip)
    Ut = ops.winograd_weights(wflip, tile=tile)
    dz = torch.tensor(_rand(rng, N, H, W, C_), device=dev)
    monkeypatch.setattr(ops, "_WINO_FUSE_MIN_TILES", 1)

    def run(fused, rep):
        monkeypatch.setattr(ops, "_WINO_FUSE", fused)
        xin = x * (1.0 + 0.25 * rep)                                  # a different problem every repetition
        z = torch.empty((N, H, W, C_), device=dev)
        y = ops.conv2d_winograd(xin, U, b, sc, sh, 1, z_out=z)
        sums = [torch.zeros(C_, device=dev) for _ in range(6)]
        d1 = ops.conv2d_dgrad_ep_winograd(dz, Ut, y, z, sc, mean, rstd, sums[0], sums[1], sums[2], 1)                  # mask from out
        d2 = ops.conv2d_dgrad_ep_winograd(dz, Ut, None, z, sc, mean, rstd, sums[3], sums[4], sums[5], 1, fwd_shift=sh)  # mask from z
        torch.cuda.synchronize()
        return y, z, d1, d2, sums
    for rep in range(3):
        want = run(False, rep)
        got = run(True, rep)
        for k in range(4):
            assert torch.equal(got[k], want[k]), (rep, k, float((got[k] - want[k]).abs().max()))
        assert torch.equal(got[2], got[3]) and float(got[2].abs().max()) > 0
        for u, v in zip(got[4], want[4]):
            torch.testing.assert_close(u, v, rtol=2e-5, atol=2e-5 * float(v.abs().max()))
    for t in ops._wino_counter_cache.values():
        assert int(t.abs().sum()) == 0                                 # every row-block counter is back at zero


def test_winograd_gemm_tiles_agree(dev, monkeypatch):
    """mrcnn_winograd_gemm: the 128 x 256 tile (taken for large products) and the 128 x 128 tile walk K in the same order, so the
    36 products of an F(4x4) layer must agree bit for bit; both against a float64 matmul of three of the matrices."""
    ops = _ops()
    lib = ops._hip.lib()
    nb, rows, K, N = 36, 640, 256, 256
    g = torch.Generator(device="cpu").manual_seed(5)
    V = torch.randn(nb, rows, K, generator=g).to(dev)
    U = (torch.randn(nb, K, N, generator=g) * 0.05).to(dev)
    outs = []
    for wide_min in ("-1", "0"):
        monkeypatch.setenv("MRCNN_WINOGRAD_GEMM_WIDE_MIN", wide_min)
        Mt = torch.full((nb, rows, N), 3.0, device=dev)
        ops.check(lib.mrcnn_winograd_gemm(ops.ptr(V), ops.ptr(U), ops.ptr(Mt), nb, rows, K, N, ops.current_stream()), "mrcnn_winograd_gemm")
        torch.cuda.synchronize()
        outs.append(Mt)
    assert torch.equal(outs[0], outs[1])
    for k in (0, 17, 35):
        ref = V[k].double().cpu() @ U[k].double().cpu()
        torch.testing.assert_close(outs[1][k].double().cpu(), ref, rtol=1e-4, atol=1e-4)


def test_conv_multi_launch_matches_oracle(dev):
    """mrcnn_conv2d_fwd_multi: the RPN model over five pyramid levels in three launches (model.py:2040-2055) --
    shared 3x3 + ReLU, then the two 1x1 heads written straight into the concatenated [B, A, *] buffers -- and four
    independent convolutions with their own weights, frozen BN and pre-BN outputs (FPN smoothing shape)."""
    ops = _ops()
    rng = np.random.default_rng(77)
    B, C, na = 2, 64, 3
    sizes = [(16, 16), (8, 8), (4, 4), (2, 2), (1, 1)]
    w = _rand(rng, 3, 3, C, 96, scale=1.0 / np.sqrt(9 * C)); b = _rand(rng, 96, scale=0.1)
    wc = _rand(rng, 1, 1, 96, 2 * na, scale=0.1); bc = _rand(rng, 2 * na, scale=0.1)
    xs = [_rand(rng, B, h, wd, C) for h, wd in sizes]
    dv = lambda a: torch.tensor(a, device=dev)
    ss = ops.conv2d_multi([dict(x=dv(x), w=dv(w), bias=dv(b), act=1) for x in xs])
    assert ss is not None
    A = sum(h * wd * na for h, wd in sizes)
    logits = torch.full((B, A, 2), -9.0, device=dev)
    off, probs, offs = 0, [], []
    for (h, wd), s in zip(sizes, ss):
        offs.append(off)
        off += h * wd * na
    ops.conv2d_multi([dict(x=s, w=dv(wc), bias=dv(bc), padding="valid", out_ptr=logits.data_ptr() + o * 2 * 4,
                           out_strides=(A * 2, s.shape[2] * 2 * na, 2 * na)) for s, o in zip(ss, offs)])
    torch.cuda.synchronize()
    ref_logits = []
    for x, s in zip(xs, ss):
        r = torch.relu(orc.conv2d_nhwc(torch.tensor(x), torch.tensor(w), torch.tensor(b), 1, "same"))
        torch.testing.assert_close(s.cpu(), r, **F32)
        ref_logits.append(orc.conv2d_nhwc(r, torch.tensor(wc), torch.tensor(bc), 1, "valid").reshape(B, -1, 2))
    torch.testing.assert_close(logits.cpu(), torch.cat(ref_logits, 1), **F32)
    # own weights per problem, frozen BN, z_out, different Cin (K-slice counts differ per problem)
    probs, refs = [], []
    for i, ((h, wd), cin) in enumerate(zip(sizes[:4], (64, 128, 32, 256))):
        x = _rand(rng, B, h, wd, cin); wi = _rand(rng, 3, 3, cin, 256, scale=1.0 / np.sqrt(9 * cin)); bi = _rand(rng, 256, scale=0.1)
        sc = rng.uniform(.5, 1.5, 256).astype(np.float32); sh = rng.uniform(-.2, .2, 256).astype(np.float32)
        z = torch.empty((B, h, wd, 256), device=dev)
        probs.append(dict(x=dv(x), w=dv(wi), bias=dv(bi), scale=dv(sc), shift=dv(sh), z_out=z))
        zr = orc.conv2d_nhwc(torch.tensor(x), torch.tensor(wi), torch.tensor(bi), 1, "same")
        refs.append((zr, zr * torch.tensor(sc) + torch.tensor(sh)))
    outs = ops.conv2d_multi(probs)
    torch.cuda.synchronize()
    for pr, o, (zr, yr) in zip(probs, outs, refs):
        torch.testing.assert_close(pr["z_out"].cpu(), zr, **F32)
        torch.testing.assert_close(o.cpu(), yr, **F32)
    # different output-tile classes do not share a launch: reported, nothing launched
    assert ops.conv2d_multi([dict(x=dv(xs[0]), w=dv(w), bias=dv(b)), dict(x=ss[0], w=dv(wc), bias=dv(bc), padding="valid")]) is None


def test_conv_into_concat_buffer(dev):
    """RPN heads write each pyramid level straight into the concatenated [B, A, 2] buffer."""
    ops = _ops()
    rng = np.random.default_rng(5)
    B = 2
    shapes = [(8, 8), (4, 4)]
    A = sum(h * w * 3 for h, w in shapes)
    buf = torch.full((B, A, 2), -7.0, device=dev)
    w = _rand(rng, 1, 1, 512, 6, scale=0.05)
    b = _rand(rng, 6, scale=0.1)
    off = 0
    refs = []
    for h, wd in shapes:
        x = _rand(rng, B, h, wd, 512)
        refs.append(orc.conv2d_nhwc(torch.tensor(x), torch.tensor(w), torch.tensor(b), 1, "valid").reshape(B, -1, 2))
        ops.conv2d_into(torch.tensor(x, device=dev), torch.tensor(w, device=dev), torch.tensor(b, device=dev),
                        buf.data_ptr() + off * 2 * 4, A * 2, wd * 6, 6, 1, "valid")
        off += h * wd * 3
    torch.cuda.synchronize()
    torch.testing.assert_close(buf.cpu(), torch.cat(refs, 1), **F32)


@pytest.mark.parametrize("case", [(3, 14, 14, 256, 256, "direct"), (3, 14, 14, 256, 256, "gemm"), (101, 14, 14, 256, 256, "gemm"),
                                  (5, 6, 10, 32, 64, "gemm")])
def test_deconv2x2(dev, case, monkeypatch):
    """Conv2DTranspose(2 x 2, stride 2) + bias + ReLU (model.py:1084-1086) against the oracle: the convolution kernel's
    pixel-shuffle store and the persistent GEMM with the same store in its epilogue (row counts that are not multiples of the
    128-row tile, a 6 x 10 map, 64 output channels = 32-column groups that change tap inside a 128-column tile)."""
    ops = _ops()
    from caesar_mrcnn_amd.params import deconv_keras_to_gemm
    N, H, W, Cin, Cd, path = case
    monkeypatch.setattr(ops, "_DECONV_GEMM", path == "gemm")
    monkeypatch.setattr(ops, "_DECONV_GEMM_MIN_ROWS", 1)
    rng = np.random.default_rng(6 + N)
    x = _rand(rng, N, H, W, Cin)
    k = _rand(rng, 2, 2, Cd, Cin, scale=0.05)      # Keras (2,2,out,in)
    b = _rand(rng, Cd, scale=0.1)
    ref = torch.relu(orc.conv2d_transpose_2x2(torch.tensor(x), torch.tensor(k), torch.tensor(b)))
    wg = torch.tensor(deconv_keras_to_gemm(k).reshape(Cin, 4 * Cd), device=dev)
    out = torch.full((N, 2 * H, 2 * W, Cd), 9.0, device=dev)
    ops.deconv2x2(torch.tensor(x, device=dev), wg, torch.tensor(b, device=dev), out=out)
    torch.cuda.synchronize()
    torch.testing.assert_close(out.cpu(), ref, **F32)
    out0 = ops.deconv2x2(torch.tensor(x, device=dev), wg, None, ops.ACT_NONE)
    torch.cuda.synchronize()
    torch.testing.assert_close(out0.cpu(), orc.conv2d_transpose_2x2(torch.tensor(x), torch.tensor(k), torch.zeros(Cd)), **F32)


WGRAD_CASES = [
    (2, 14, 14, 256, 256, 3, 1, "same"),
    (2, 16, 16, 64, 256, 1, 1, "valid"),
    (2, 16, 16, 256, 128, 1, 2, "valid"),
    (2, 32, 32, 3, 64, 7, 2, (3, 3)),
    (6, 7, 7, 256, 1024, 7, 1, "valid"),
    (40, 14, 14, 256, 256, 3, 1, "same"),       # several pixel splits
    (2, 16, 16, 512, 6, 1, 1, "valid"),
    (2, 16, 16, 64, 64, 3, 1, "same"),
    (3, 9, 9, 128, 64, 1, 1, "valid"),
    (2, 16, 16, 1024, 512, 1, 2, "valid"),      # strided-scatter dgrad through the split-K path
    (2, 8, 8, 2048, 512, 1, 1, "valid"),
    (3, 13, 11, 128, 256, 3, 1, "same"),        # LDS-DMA wgrad: M = 429 (pixel tail inside a 16-row piece), odd H/W
    (90, 13, 11, 128, 256, 3, 1, "same"),       # LDS-DMA wgrad + LDS-DMA dgrad, ragged M
    (459, 13, 11, 128, 128, 3, 1, "same"),      # M = 65637 >= 65536: pixel-table addressing, tail inside a step
]


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_conv_wgrad_and_dgrad(dev, case):
    ops = _ops()
    N, H, W, Cin, Cout, k, stride, padding = case
    rng = np.random.default_rng(1000 + sum(int(v) if isinstance(v, int) else 7 for v in case))
    x = torch.tensor(_rand(rng, N, H, W, Cin), requires_grad=True)
    w = torch.tensor(_rand(rng, k, k, Cin, Cout, scale=1.0 / np.sqrt(k * k * Cin)), requires_grad=True)
    y = orc.conv2d_nhwc(x, w, None, stride, padding)
    dy = torch.tensor(_rand(rng, *y.shape))
    y.backward(dy)
    xt, dyt = x.detach().to(dev), dy.to(dev)
    dw = ops.conv2d_wgrad(xt, dyt, tuple(w.shape), stride, padding)
    torch.cuda.synchronize()
    torch.testing.assert_close(dw.cpu(), w.grad, rtol=5e-4, atol=5e-4 * float(w.grad.abs().max()))
    # accumulate mode
    dw2 = ops.conv2d_wgrad(xt, dyt, tuple(w.shape), stride, padding, dw=dw.clone(), accumulate=True)
    torch.testing.assert_close(dw2.cpu(), 2 * w.grad, rtol=5e-4, atol=1e-3 * float(w.grad.abs().max()))
    # data gradient = conv of dy with the flipped/transposed weights (stride 1) or strided scatter (1x1 s2)
    wt = ops.weight_flip_transpose(w.detach().to(dev))
    if stride == 1 and k > 1 and padding == "same":
        dx = ops.conv2d(dyt, wt, None, None, None, None, 1, ((k - 1) // 2, (k - 1) // 2))
        torch.testing.assert_close(dx.cpu(), x.grad, **F32)
    elif k == 1 and stride == 1:
        dx = ops.conv2d(dyt, wt, None, None, None, None, 1, "valid")
        torch.testing.assert_close(dx.cpu(), x.grad, **F32)
    elif k == 1 and stride == 2:
        dx = torch.zeros((N, H, W, Cin), device=dev)
        d = ops.conv_desc(tuple(dyt.shape), tuple(wt.shape), 1, "valid")
        d.out_w_stride, d.out_h_stride, d.out_n_stride = 2 * Cin, 2 * W * Cin, H * W * Cin
        ops.conv2d(dyt, wt, out=dx, desc=d)
        torch.testing.assert_close(dx.cpu(), x.grad, **F32)


def test_conv_wgrad_multi_launch(dev):
    """mrcnn_conv2d_wgrad_multi: the weight gradients of a bottleneck block (1x1, 3x3, 1x1, and the stride-2 1x1 shortcut
    of a stage's first block) in one launch, plain and accumulating; a layer that does not fit (Cout = 64) is refused."""
    ops = _ops()
    rng = np.random.default_rng(4242)
    N, H, W = 3, 9, 7                                   # M = 189: pixel tail inside a 16-row piece
    layers = [(256, 128, 1, 1, "valid", H, W), (128, 128, 3, 1, "same", H, W), (128, 512, 1, 1, "valid", H, W),
              (256, 512, 1, 2, "valid", 2 * H, 2 * W)]
    items, refs = [], []
    for cin, cout, k, stride, padding, h, w_ in layers:
        x = torch.tensor(_rand(rng, N, h, w_, cin))
        w = torch.tensor(_rand(rng, k, k, cin, cout, scale=1.0 / np.sqrt(k * k * cin)), requires_grad=True)
        y = orc.conv2d_nhwc(x, w, None, stride, padding)
        dy = torch.tensor(_rand(rng, *y.shape))
        y.backward(dy)
        refs.append(w.grad)
        items.append((x.to(dev), dy.to(dev), tuple(w.shape), stride, padding, torch.full(tuple(w.shape), 0.5, device=dev), False))
    assert ops.conv2d_wgrad_multi(items)
    torch.cuda.synchronize()
    for it, r in zip(items, refs):
        torch.testing.assert_close(it[5].cpu(), r, rtol=5e-4, atol=5e-4 * float(r.abs().max()))
    acc = [it[:6] + (True,) for it in items]
    assert ops.conv2d_wgrad_multi(acc)
    torch.cuda.synchronize()
    for it, r in zip(items, refs):
        torch.testing.assert_close(it[5].cpu(), 2 * r, rtol=5e-4, atol=1e-3 * float(r.abs().max()))
    x = torch.tensor(_rand(rng, N, H, W, 256), device=dev); dy = torch.tensor(_rand(rng, N, H, W, 64), device=dev)
    assert not ops.conv2d_wgrad_multi([items[0], (x, dy, (1, 1, 256, 64), 1, "valid", torch.zeros(1, 1, 256, 64, device=dev), False)])


@pytest.mark.parametrize("M,C,act,bn", [(500, 256, 1, True), (1000, 6, 0, False), (3000, 4, 2, False),
                                         (77, 2048, 1, True), (4096, 64, 1, True)])
def test_epilogue_bwd(dev, M, C, act, bn):
    ops = _ops()
    rng = np.random.default_rng(M + C)
    z = torch.tensor(_rand(rng, M, C), requires_grad=True)
    g, be, mu, var = (torch.tensor(rng.uniform(.5, 1.5, C).astype(np.float32), requires_grad=True),
                      torch.tensor(rng.uniform(-.2, .2, C).astype(np.float32), requires_grad=True),
                      torch.tensor(rng.uniform(-.2, .2, C).astype(np.float32)),
                      torch.tensor(rng.uniform(.5, 1.5, C).astype(np.float32)))
    y = orc.batchnorm_frozen(z, g, be, mu, var) if bn else z
    out = torch.relu(y) if act == 1 else (torch.sigmoid(y) if act == 2 else y)
    dout = torch.tensor(_rand(rng, M, C))
    out.backward(dout)
    rstd = 1.0 / torch.sqrt(var + orc.BN_EPS)
    scale = (g * rstd).detach()
    D = lambda t: None if t is None else t.detach().to(dev).contiguous()
    dy_o, dz_o = torch.empty(M, C, device=dev), torch.empty(M, C, device=dev)
    dgam, dbet, dbias = (torch.zeros(C, device=dev) for _ in range(3))
    ops.epilogue_bwd(D(dout), D(out), D(z) if bn else None, D(scale) if bn else None, D(mu) if bn else None,
                     D(rstd) if bn else None, dy_o, dz_o, dgam if bn else None, dbet if bn else None, dbias, act)
    torch.cuda.synchronize()
    torch.testing.assert_close(dz_o.cpu(), z.grad, rtol=1e-5, atol=1e-6)
    tol = dict(rtol=1e-3, atol=1e-3 * max(1.0, float(z.grad.abs().sum(0).max())))
    torch.testing.assert_close(dbias.cpu(), z.grad.sum(0), **tol)
    if bn:
        torch.testing.assert_close(dgam.cpu(), g.grad, **tol)
        torch.testing.assert_close(dbet.cpu(), be.grad, **tol)


def test_pool_and_resample(dev):
    ops = _ops()
    rng = np.random.default_rng(11)
    for (N, H, W, C) in [(2, 32, 32, 64), (1, 17, 23, 8)]:
        x = torch.tensor(_rand(rng, N, H, W, C), requires_grad=True)
        ref = orc.maxpool3x3s2_same(x)
        dy = torch.tensor(_rand(rng, *ref.shape))
        ref.backward(dy)
        out, am = ops.maxpool3x3s2(x.detach().to(dev), want_argmax=True)
        dx = ops.maxpool3x3s2_bwd(dy.to(dev), am, tuple(x.shape))
        torch.cuda.synchronize()
        assert torch.equal(out.cpu(), ref.detach())
        torch.testing.assert_close(dx.cpu(), x.grad, rtol=1e-6, atol=1e-6)
    x = torch.tensor(_rand(rng, 2, 8, 8, 256))
    assert torch.equal(ops.subsample2(x.to(dev)).cpu(), x[:, ::2, ::2, :])
    dx = torch.ones(2, 8, 8, 256, device=dev)
    d6 = torch.tensor(_rand(rng, 2, 4, 4, 256))
    ops.subsample2_bwd_acc(d6.to(dev), dx)
    ref = torch.ones(2, 8, 8, 256); ref[:, ::2, ::2, :] += d6
    assert torch.equal(dx.cpu(), ref)
    dup = torch.tensor(_rand(rng, 2, 16, 16, 32))
    dsrc = torch.ones(2, 8, 8, 32, device=dev)
    ops.upsample2_bwd(dup.to(dev), dsrc, True)
    ref = 1 + dup.reshape(2, 8, 2, 8, 2, 32).sum((2, 4))
    torch.testing.assert_close(dsrc.cpu(), ref, rtol=1e-6, atol=1e-6)
    lg = torch.tensor(_rand(rng, 1000, 4, scale=3))
    torch.testing.assert_close(ops.softmax_rows(lg.to(dev)).cpu(), torch.softmax(lg, -1), rtol=1e-6, atol=1e-7)
    a = torch.tensor(_rand(rng, 100000)); b = torch.tensor(_rand(rng, 100000))
    ad = a.to(dev); ops.add_inplace(ad, b.to(dev))
    assert torch.equal(ad.cpu(), a + b)


def _random_rois(rng, B, R, zero_tail=0):
    """log-uniform sizes covering all four pyramid levels, some ROIs touching the border."""
    size = np.exp(rng.uniform(np.log(0.01), np.log(0.9), (B, R, 2)))
    ctr = rng.uniform(0, 1, (B, R, 2))
    y1 = np.clip(ctr[..., 0] - size[..., 0] / 2, 0, 1); y2 = np.clip(ctr[..., 0] + size[..., 0] / 2, 0, 1)
    x1 = np.clip(ctr[..., 1] - size[..., 1] / 2, 0, 1); x2 = np.clip(ctr[..., 1] + size[..., 1] / 2, 0, 1)
    rois = np.stack([y1, x1, y2, x2], -1).astype(np.float32)
    if zero_tail:
        rois[:, -zero_tail:] = 0
    return rois


@pytest.mark.parametrize("pool", [7, 14])
def test_roialign_fwd_bwd(dev, pool):
    ops = _ops()
    rng = np.random.default_rng(pool)
    B, R, C = 2, 60, 256
    fms = [torch.tensor(_rand(rng, B, s, s, C), requires_grad=True) for s in (64, 32, 16, 8)]
    rois = _random_rois(rng, B, R, zero_tail=3)
    rois[0, 0] = [0.2, 0.2, 1.2, 0.7]       # partly outside -> extrapolation zeros
    ref = orc.pyramid_roi_align(rois, fms, pool, 1024.0 * 1024.0)
    dout = torch.tensor(_rand(rng, *ref.shape))
    ref.backward(dout)
    fd = [f.detach().to(dev) for f in fms]
    out, lv = ops.roialign(torch.tensor(rois, device=dev), fd, pool, 1024.0 * 1024.0, want_levels=True)
    torch.cuda.synchronize()
    assert torch.equal(lv.cpu().long(), orc.roi_levels(rois, 1024.0 * 1024.0))
    assert len(set(lv.cpu().numpy().ravel().tolist())) == 4
    torch.testing.assert_close(out.cpu(), ref.detach(), rtol=1e-5, atol=1e-5)
    dfm = [torch.zeros_like(f) for f in fd]
    ops.roialign_bwd(torch.tensor(rois, device=dev), dout.to(dev), dfm, pool, 1024.0 * 1024.0)
    torch.cuda.synchronize()
    for a, f in zip(dfm, fms):
        torch.testing.assert_close(a.cpu(), f.grad, rtol=1e-4, atol=1e-4)
    # gather form (one wave per destination pixel): same adjoint, accumulating onto what the maps already hold
    dfm2 = [torch.full_like(f, 0.25) for f in fd]
    ops.roialign_bwd(torch.tensor(rois, device=dev), dout.to(dev), dfm2, pool, 1024.0 * 1024.0, dense=True)
    torch.cuda.synchronize()
    for a, f in zip(dfm2, fms):
        torch.testing.assert_close(a.cpu() - 0.25, f.grad, rtol=1e-4, atol=1e-4)


def test_roialign_bwd_gather_heavy_overlap(dev):
    """512 ROIs per image clustered on a few objects (the class head's training shape at 256^2 inputs: hundreds of
    contributions per pixel, zero-padded rows all on pixel (0,0))."""
    ops = _ops()
    rng = np.random.default_rng(99)
    B, R, C, pool = 2, 512, 256, 7
    fms = [torch.zeros(B, s, s, C, requires_grad=True) for s in (64, 32, 16, 8)]
    ctr = rng.uniform(0.3, 0.7, (B, 4, 2))
    rois = np.zeros((B, R, 4), np.float32)
    for b in range(B):
        for r in range(R - 20):
            c = ctr[b, r % 4] + rng.normal(0, 0.03, 2)
            hw = np.exp(rng.uniform(np.log(0.08), np.log(0.9), 2))
            rois[b, r] = np.clip([c[0] - hw[0] / 2, c[1] - hw[1] / 2, c[0] + hw[0] / 2, c[1] + hw[1] / 2], 0, 1)
    area = 256.0 * 256.0
    ref = orc.pyramid_roi_align(rois, fms, pool, area)
    dout = torch.tensor(_rand(rng, *ref.shape))
    ref.backward(dout)
    dfm = [torch.zeros(B, s, s, C, device=dev) for s in (64, 32, 16, 8)]
    ops.roialign_bwd(torch.tensor(rois, device=dev), dout.to(dev), dfm, pool, area, dense=True)
    torch.cuda.synchronize()
    for a, f in zip(dfm, fms):
        g = f.grad if f.grad is not None else torch.zeros_like(f)          # no ROI reaches P5 at this image area
        torch.testing.assert_close(a.cpu(), g, rtol=2e-4, atol=2e-3)          # sums of several hundred terms per pixel


@pytest.mark.parametrize("A,limit,count,thr", [(16368, 6000, 2000, 0.7), (16368, 6000, 1000, 0.7), (3000, 6000, 1000, 0.5),
                                                (65472, 6000, 1000, 0.9)])
def test_proposal_layer(dev, A, limit, count, thr):
    ops = _ops()
    rng = np.random.default_rng(A + count)
    B = 2
    cfg = _cfg()
    cfg.PRE_NMS_LIMIT, cfg.RPN_NMS_THRESHOLD = limit, thr
    ctr = rng.uniform(0, 1, (A, 2)); sz = np.exp(rng.uniform(np.log(0.02), np.log(0.5), (A, 2)))
    anchors = np.concatenate([ctr - sz / 2, ctr + sz / 2], 1).astype(np.float32)
    fg = 1 / (1 + np.exp(-rng.normal(0, 2, (B, A))))
    fg = np.round(fg, 3)                                  # many exact score ties -> exercises the tie rule
    probs = np.stack([1 - fg, fg], -1).astype(np.float32)
    deltas = _rand(rng, B, A, 4, scale=0.5)
    o = orc.OracleMaskRCNN(cfg, {})
    ref_rois, det = o.proposal_layer(torch.tensor(probs), torch.tensor(deltas), anchors, count, detail=True)
    rois, top_idx, keep_idx, num_keep, boxes = ops.proposals(
        torch.tensor(probs, device=dev), torch.tensor(deltas, device=dev), torch.tensor(anchors, device=dev),
        limit, count, thr, cfg.RPN_BBOX_STD_DEV, debug=True)
    torch.cuda.synchronize()
    for b in range(B):
        ix, ref_boxes, keep = det[b]
        assert np.array_equal(top_idx[b].cpu().numpy(), ix.astype(np.int32)), "top-k order differs"
        gb = boxes[b].cpu().numpy()
        np.testing.assert_allclose(gb, ref_boxes, rtol=0, atol=2e-6)   # expf vs np.exp: a few ulp
        # NMS is checked index-exact on identical inputs: the oracle re-runs on the GPU-decoded boxes
        keep2 = orc.tf_non_max_suppression(gb, probs[b, ix, 1], count, thr)
        n = int(num_keep[b])
        assert n == len(keep2)
        assert np.array_equal(keep_idx[b, :n].cpu().numpy(), keep2.astype(np.int32))
        assert np.all(keep_idx[b, n:].cpu().numpy() == -1)
        np.testing.assert_array_equal(rois[b, :n].cpu().numpy(), gb[keep2])
        assert np.all(rois[b, n:].cpu().numpy() == 0)


def test_proposal_nms_at_exact_threshold_pairs(dev):
    """The reference-generated fixture with pairs whose IoU EQUALS the threshold (0.5, 0.25; exact in float32) through
    the HIP proposal layer: zero deltas decode to the anchors themselves (coordinates are multiples of 1/128, every
    step exact), so the kernel's keep list must be the reference NumPy NMS's -- both boxes survive at equality."""
    ops = _ops()
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_numpy_helpers.npz"))
    eb, es = G["nms_eq_boxes"], G["nms_eq_scores"]
    n = eb.shape[0]
    anchors = torch.tensor((eb / np.float32(128)).astype(np.float32), device=dev)
    probs = torch.tensor(np.stack([1 - es, es], -1)[None].astype(np.float32), device=dev)
    deltas = torch.zeros((1, n, 4), device=dev)
    for thr, tag in ((0.5, "50"), (0.25, "25"), (0.6, "60")):
        rois, top_idx, keep_idx, num_keep, boxes = ops.proposals(probs, deltas, anchors, 6000, n, thr, _cfg().RPN_BBOX_STD_DEV,
                                                                 debug=True)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(boxes[0].cpu().numpy(), eb / np.float32(128))
        k = int(num_keep[0])
        kept = top_idx[0].cpu().numpy()[keep_idx[0, :k].cpu().numpy()]
        np.testing.assert_array_equal(kept, G["nms_eq_keep_" + tag])


def test_proposal_selection_survives_stale_counters(dev):
    """Fault injection for the multi-workgroup top-k (A >= 32 768): the second call runs with the reset of its
    histograms / counters suppressed (mrcnn_tuning_set("proposal_skip_zero", 1)), i.e. with the state a missing reset would leave.
    The slot taken from the global counter is bounds-guarded (nothing is written outside the candidate array, refused
    stores are counted) and the sort kernel, seeing collected != announced, selects for itself: same indices as the
    healthy call.  Regression for the 1024x1024 graph-replay memory fault of round 1."""
    import os
    ops = _ops()
    rng = np.random.default_rng(77)
    B, A, limit, count, thr = 2, 65472, 6000, 1000, 0.7
    cfg = _cfg()
    ctr = rng.uniform(0, 1, (A, 2)); sz = np.exp(rng.uniform(np.log(0.02), np.log(0.5), (A, 2)))
    anchors = torch.tensor(np.concatenate([ctr - sz / 2, ctr + sz / 2], 1).astype(np.float32), device=dev)
    fg = np.round(1 / (1 + np.exp(-rng.normal(0, 2, (B, A)))), 3)
    probs = torch.tensor(np.stack([1 - fg, fg], -1).astype(np.float32), device=dev)
    deltas = torch.tensor(_rand(rng, B, A, 4, scale=0.5), device=dev)
    args = (probs, deltas, anchors, limit, count, thr, cfg.RPN_BBOX_STD_DEV)
    good = [t.cpu().numpy().copy() for t in ops.proposals(*args, debug=True)]
    st = ops.proposal_status(probs, limit, count)
    assert (st[:, 0] == limit).all() and (st[:, 1] == limit).all() and (st[:, 2] == 0).all(), st
    ops.tuning_set("proposal_skip_zero", 1)
    try:
        for rep in range(3):                              # counters keep growing: 2K, 3K, 4K > SORT_CAP
            bad = [t.cpu().numpy().copy() for t in ops.proposals(*args, debug=True)]
            st = ops.proposal_status(probs, limit, count)
            assert (st[:, 0] != limit).all(), st         # the pre-selection is visibly inconsistent ...
            for g, b_ in zip(good, bad):                  # ... and the result is still the exact one
                assert np.array_equal(g, b_), rep
        assert (st[:, 2] > 0).any(), st                   # by now some stores were refused by the guard
    finally:
        ops.tuning_set("proposal_skip_zero", 0)
    again = [t.cpu().numpy().copy() for t in ops.proposals(*args, debug=True)]
    st = ops.proposal_status(probs, limit, count)
    assert (st[:, 0] == limit).all() and (st[:, 2] == 0).all()
    for g, b_ in zip(good, again):
        assert np.array_equal(g, b_)


def test_detection_layer(dev):
    ops = _ops()
    rng = np.random.default_rng(21)
    cfg = _cfg(mode="inference")
    B, R, C = 2, 1000, cfg.NUM_CLASSES
    rois = _random_rois(rng, B, R, zero_tail=50)
    lg = _rand(rng, B, R, C, scale=2.0)
    probs = np.exp(lg) / np.exp(lg).sum(-1, keepdims=True)
    probs = np.round(probs, 2).astype(np.float32)            # score ties
    deltas = _rand(rng, B, R, C, 4, scale=0.5)
    windows = np.array([[0, 0, 1, 1], [0.1, 0.05, 0.9, 0.95]], np.float32)
    o = orc.OracleMaskRCNN(cfg, {})
    for minconf in (0, 0.5):
        cfg.DETECTION_MIN_CONFIDENCE = minconf
        out = ops.detections(torch.tensor(rois, device=dev), torch.tensor(probs, device=dev),
                             torch.tensor(deltas, device=dev), torch.tensor(windows, device=dev),
                             cfg.DETECTION_MAX_INSTANCES, minconf, cfg.DETECTION_NMS_THRESHOLD, cfg.BBOX_STD_DEV)
        torch.cuda.synchronize()
        for b in range(B):
            ref = o.refine_detections(rois[b], probs[b], deltas[b], windows[b])
            got = out[b].cpu().numpy()
            assert np.array_equal(got[:, 4], ref[:, 4]), "class ids / order differ"
            assert np.array_equal(got[:, 5], ref[:, 5])
            np.testing.assert_allclose(got[:, :4], ref[:, :4], rtol=0, atol=2e-6)
            assert (ref[:, 4] > 0).sum() > 10


@pytest.mark.parametrize("mini", [False, True])
def test_detection_targets(dev, mini):
    """mini=True: config.USE_MINI_MASK -- gt_masks hold each instance cropped to its box and resized to MINI_MASK_SHAPE
    (56x56) and the ROI is re-expressed in the GT box's frame before crop_and_resize (mrcnn/model.py:670-682)."""
    ops = _ops()
    rng = np.random.default_rng(31)
    cfg = _cfg()
    cfg.USE_MINI_MASK = mini
    B, R, G, T, HW = 2, 2000, cfg.MAX_GT_INSTANCES, cfg.TRAIN_ROIS_PER_IMAGE, 64
    gt_boxes = np.zeros((B, G, 4), np.float32); gt_cls = np.zeros((B, G), np.int32)
    gt_masks = np.zeros((B, HW, HW, G), np.uint8)
    props = _random_rois(rng, B, R, zero_tail=300)
    for b in range(B):
        ng = 7 + b
        for g in range(ng):
            y1, x1 = rng.integers(0, HW - 12, 2); h, w = rng.integers(4, 12, 2)
            gt_masks[b, y1:y1 + h, x1:x1 + w, g] = 1
            gt_masks[b, y1, x1, g] = 0
            box = np.array([y1, x1, y1 + h, x1 + w], np.float32)
            gt_boxes[b, g] = (box - [0, 0, 1, 1]) / (HW - 1)
            gt_cls[b, g] = rng.integers(1, cfg.NUM_CLASSES)
            # jittered proposals around every GT so there are positives
            for j in range(40):
                props[b, g * 40 + j] = np.clip(gt_boxes[b, g] + rng.normal(0, 0.01, 4), 0, 1)
        gt_cls[b, ng - 1] = -1           # one crowd box
    props[0, 5] = 0                      # a zero row in the middle
    keys = rng.uniform(0, 1, (B, R)).astype(np.float32)
    o = orc.OracleMaskRCNN(cfg, {})
    if mini:                     # any binary planes do: the layer only samples them in the GT box's frame
        gt_masks = (rng.uniform(0, 1, (B, 56, 56, G)) < 0.6).astype(np.uint8)
    got = ops.detection_targets(*(torch.tensor(a, device=dev) for a in (props, gt_cls, gt_boxes, gt_masks, keys)),
                                T, cfg.ROI_POSITIVE_RATIO, cfg.BBOX_STD_DEV, cfg.MASK_SHAPE, use_mini_mask=mini)
    torch.cuda.synchronize()
    rois, tcls, tbbox, tmask, assign, counts = (t.cpu().numpy() for t in got)
    for b in range(B):
        r_rois, r_cls, r_bb, r_m, (P, N) = o.detection_targets(props[b], gt_cls[b], gt_boxes[b], gt_masks[b].astype(bool), keys[b])
        assert (counts[b, 0], counts[b, 1]) == (P, N) and P > 20 and N > 20
        np.testing.assert_array_equal(rois[b], r_rois)
        np.testing.assert_array_equal(tcls[b], r_cls)
        np.testing.assert_allclose(tbbox[b], r_bb, rtol=1e-5, atol=1e-5)
        # byte work: bit-exact.  The kernel evaluates tf.image.crop_and_resize's float32 expressions in TF's operation
        # order without contraction and rounds half to even like tf.round (rintf); the oracle does the same in torch
        np.testing.assert_array_equal(tmask[b], r_m)
        assert r_m.sum() > 0 and (r_m[:P] == 0).sum() > 0


@pytest.mark.parametrize("case", ["golden", "crowded", "big_image", "no_overlap_gt"])
def test_rpn_targets_device(dev, case):
    """mrcnn_rpn_targets vs the oracle's build_rpn_targets (model.py:1536-1644) with the keyed draw:
    rpn_match bit-exact (float64 IoUs in the reference's operation order), box deltas to float32 rounding."""
    import os
    ops = _ops()
    cfg = _cfg()
    rng = np.random.default_rng({"golden": 3, "crowded": 5, "big_image": 8, "no_overlap_gt": 13}[case])
    S = 1024 if case == "big_image" else 256
    anchors = orc.generate_pyramid_anchors(cfg.RPN_ANCHOR_SCALES, cfg.RPN_ANCHOR_RATIOS,
                                           orc.compute_backbone_shapes(cfg.BACKBONE_STRIDES, (S, S)),
                                           cfg.BACKBONE_STRIDES, cfg.RPN_ANCHOR_STRIDE)
    assert anchors.dtype == np.float64
    A, G, NT = anchors.shape[0], cfg.MAX_GT_INSTANCES, cfg.RPN_TRAIN_ANCHORS_PER_IMAGE
    B = 3
    gt_cls = np.zeros((B, G), np.int32); gt_boxes = np.zeros((B, G, 4), np.int32)
    if case == "golden":
        gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_numpy_helpers.npz"))
        n = gold["rpn_gt_ids"].shape[0]
        gt_cls[0, :n], gt_boxes[0, :n] = gold["rpn_gt_ids"], gold["rpn_gt_boxes"]
        gt_cls[1, :n], gt_boxes[1, :n] = gold["rpn_gt_ids_crowd"], gold["rpn_gt_boxes"]
        gt_cls[2, 5:5 + n], gt_boxes[2, 5:5 + n] = gold["rpn_gt_ids"], gold["rpn_gt_boxes"]     # padding rows first
    else:
        for b in range(B):
            n = {"crowded": (150, 300, 40), "big_image": (6, 1, 30), "no_overlap_gt": (4, 9, 2)}[case][b]
            for g in range(n):
                h, w = rng.integers(2, 90 if case != "crowded" else 40, 2)
                y1, x1 = rng.integers(0, S - h), rng.integers(0, S - w)
                gt_boxes[b, g] = (y1, x1, y1 + h, x1 + w)
                gt_cls[b, g] = rng.integers(1, cfg.NUM_CLASSES)
            if case == "crowded" and b == 0:
                gt_cls[b, 3] = -2; gt_cls[b, 77] = -1
            if case == "no_overlap_gt":
                gt_boxes[b, 0] = (10, 10, 10, 30)        # zero-area GT: column of zero IoUs (reference quirk: every
                                                         # zero-overlap anchor ties for "best" and turns positive)
    keys = rng.uniform(0, 1, (B, A)).astype(np.float32)
    keys[0, :200] = keys[0, 200]                         # exercise the tie rule
    std = np.asarray(cfg.RPN_BBOX_STD_DEV, np.float64)
    match, bbox = ops.rpn_targets(torch.tensor(anchors, device=dev), torch.tensor(gt_cls, device=dev),
                                  torch.tensor(gt_boxes, device=dev), torch.tensor(keys, device=dev), NT, std)
    torch.cuda.synchronize()
    match, bbox = match.cpu().numpy(), bbox.cpu().numpy()
    assert match.shape == (B, A, 1) and bbox.shape == (B, NT, 4)
    seen_pos_cap = False
    for b in range(B):
        nz = gt_cls[b] != 0
        rm, rb = orc.build_rpn_targets(anchors, gt_cls[b][nz], gt_boxes[b][nz], NT, std, rng=orc.KeyedChoice(keys[b]))
        np.testing.assert_array_equal(match[b, :, 0], rm)
        np.testing.assert_allclose(bbox[b], rb.astype(np.float32), rtol=2e-6, atol=1e-6)
        assert (rm == 1).sum() > 0 and (rm == 1).sum() + (rm == -1).sum() <= NT
        seen_pos_cap |= (rm == 1).sum() == NT // 2
    if case in ("crowded", "no_overlap_gt"):
        assert seen_pos_cap                               # the positive cap (and its keyed draw) was exercised
    if case == "golden":                                  # no positive surplus there: positives / deltas == the reference's own
        assert np.array_equal(match[0, :, 0] == 1, gold["rpn_match"] == 1)
        assert (match[0, :, 0] == -1).sum() == (gold["rpn_match"] == -1).sum()
        np.testing.assert_allclose(bbox[0], gold["rpn_bbox"].astype(np.float32), rtol=2e-6, atol=1e-6)
        assert np.array_equal(match[1, :, 0] == 1, gold["rpn_match_crowd"] == 1)
        np.testing.assert_array_equal(bbox[2], bbox[0])


@pytest.mark.parametrize("dice", [False, True])
def test_losses(dev, dice):
    ops = _ops()
    rng = np.random.default_rng(41)
    cfg = _cfg()
    cfg.MASK_LOSS_FUNCTION = "dice_coef_loss" if dice else "binary_crossentropy"
    B, A, T, C = 2, 4092, 64, cfg.NUM_CLASSES
    rpn_match = rng.choice([-1, 0, 1], size=(B, A, 1), p=[0.1, 0.85, 0.05]).astype(np.int32)
    maxpos = int((rpn_match == 1).sum(1).max()) + 5
    rpn_bbox_t = _rand(rng, B, maxpos, 4)
    rpn_logits = torch.tensor(_rand(rng, B, A, 2, scale=2), requires_grad=True)
    rpn_bbox = torch.tensor(_rand(rng, B, A, 4, scale=1.5), requires_grad=True)
    tcls = np.zeros((B, T), np.int32); tcls[:, :20] = rng.integers(1, C, (B, 20))
    tbbox = _rand(rng, B, T, 4); tmask = (rng.uniform(0, 1, (B, T, 28, 28)) > 0.5).astype(np.float32)
    active = np.ones((B, C), np.int32); active[:, 2] = 0
    logits = torch.tensor(_rand(rng, B, T, C, scale=2), requires_grad=True)
    mbbox = torch.tensor(_rand(rng, B, T, C, 4, scale=1.5), requires_grad=True)
    mm = rng.uniform(0, 1, (B, T, 28, 28, C)).astype(np.float32); mm[0, 0, 0, :4, :] = [[0.0] * C, [1.0] * C, [1e-9] * C, [0.5] * C]
    mmask = torch.tensor(mm, requires_grad=True)
    w = [1.0, 0.5, 1.0, 2.0, 1.5]
    o = orc.OracleMaskRCNN(cfg, {})
    ls = o.losses(rpn_match, rpn_bbox_t, rpn_logits, rpn_bbox, tcls, tbbox, tmask, active, logits, mbbox, mmask)
    sum(wi * li for wi, li in zip(w, ls)).backward()
    D = lambda a: torch.tensor(np.ascontiguousarray(a), device=dev) if isinstance(a, np.ndarray) else a.detach().to(dev)
    out = ops.losses_fwd_bwd(D(rpn_match), D(rpn_bbox_t), D(rpn_logits), D(rpn_bbox), D(tcls), D(tbbox), D(tmask),
                             D(active[0].copy()), D(logits), D(mbbox), D(mmask), w, dice)
    torch.cuda.synchronize()
    np.testing.assert_allclose(out[0].cpu().numpy(), [float(l) for l in ls], rtol=2e-5, atol=1e-6)
    for got, ref in zip(out[1:], (rpn_logits, rpn_bbox, logits, mbbox, mmask)):
        torch.testing.assert_close(got.cpu(), ref.grad, rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("n_used,G", [(0, 12), (1, 12), (7, 12), (8, 8), (13, 300), (300, 300)])
def test_unpack_mask_bits(dev, n_used, G):
    """mrcnn_unpack_mask_bits: GT instance masks cross PCIe as numpy.packbits(..., bitorder="little") along the instance
    axis; the device writes the uint8 planes the target kernels read, zeros for the padding instances.  Bit-exact."""
    ops = _ops()
    rng = np.random.default_rng(60 + n_used)
    B, H, W = 2, 9, 7
    masks = np.zeros((B, H, W, G), np.uint8)
    masks[..., :n_used] = rng.integers(0, 2, (B, H, W, n_used))
    if n_used == 0:                                   # nothing to read: all planes are padding
        got = ops.unpack_mask_bits(torch.zeros((B, H, W, 1), dtype=torch.uint8, device=dev), 0, G)
        torch.cuda.synchronize()
        assert tuple(got.shape) == (B, H, W, G) and int(got.sum()) == 0
        return
    packed = np.packbits(masks[..., :n_used] != 0, axis=-1, bitorder="little")
    assert packed.shape[-1] == (n_used + 7) // 8
    got = ops.unpack_mask_bits(torch.tensor(packed, device=dev), n_used, G)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(got.cpu().numpy(), masks)


def _unmold_boxes_for_test(rng, n, H, W):
    """Random integer boxes inside an H x W image: 1-pixel boxes, 1-pixel-wide rows / columns, boxes touching every edge,
    the full tile, sizes below / equal to / above the 28-pixel mask (down-, iso- and up-sampling)."""
    boxes = np.zeros((n, 4), np.int32)
    for i in range(n):
        kind = i % 10
        if kind == 0:                                   # one pixel
            y1, x1 = rng.integers(0, H), rng.integers(0, W); y2, x2 = y1 + 1, x1 + 1
        elif kind == 1:                                 # the whole tile
            y1, x1, y2, x2 = 0, 0, H, W
        elif kind == 2:                                 # touches the top-left corner
            y1, x1 = 0, 0; y2, x2 = rng.integers(1, H + 1), rng.integers(1, W + 1)
        elif kind == 3:                                 # touches the bottom-right corner
            y2, x2 = H, W; y1, x1 = rng.integers(0, H), rng.integers(0, W)
        elif kind == 4:                                 # one pixel wide / high
            y1 = rng.integers(0, H); y2 = y1 + 1; x1 = rng.integers(0, W - 1); x2 = rng.integers(x1 + 1, W + 1)
        elif kind == 5:
            x1 = rng.integers(0, W); x2 = x1 + 1; y1 = rng.integers(0, H - 1); y2 = rng.integers(y1 + 1, H + 1)
        elif kind == 6:                                 # exactly the mask's size
            y1, x1 = rng.integers(0, H - 27), rng.integers(0, W - 27); y2, x2 = y1 + 28, x1 + 28
        else:
            y1, x1 = rng.integers(0, H - 1), rng.integers(0, W - 1)
            y2, x2 = rng.integers(y1 + 1, H + 1), rng.integers(x1 + 1, W + 1)
        boxes[i] = (y1, x1, y2, x2)
    return boxes


@pytest.mark.parametrize("packed", [False, True])
def test_unmold_masks(dev, packed):
    """mrcnn_unmold_masks against the host statement of mrcnn/utils.py:629-645 (caesar_mrcnn_amd.utils.unmold_mask: float64
    bilinear resize of the 28 x 28 class mask to the integer box, clip to the mask's range, `>= 0.5`, paste): 1040 boxes in
    batches of 104 detections (13 bytes in the packed form, the last one partial) on a 96 x 80 image -- 1-pixel, edge-touching,
    full-tile, down- / up-sampled boxes; masks with all values above / below the threshold (the clip decides those) and one
    with a NaN.  Boolean output identical, both layouts; class channel and row indirection exercised."""
    ops = _ops()
    from caesar_mrcnn_amd import utils
    rng = np.random.default_rng(77)
    H, W, C_, R, n = 96, 80, 4, 120, 104
    for batch in range(10):
        mm = 1.0 / (1.0 + np.exp(-_rand(rng, R, 28, 28, C_, scale=2.0)))
        mm = mm.astype(np.float32)
        mm[3] = 0.5 + 0.4 * rng.random((28, 28, C_), dtype=np.float32)       # min >= 0.5: the clip fills the whole box
        mm[4] = 0.4 * rng.random((28, 28, C_), dtype=np.float32)             # max < 0.5: nothing set
        mm[5, 7, 9, :] = np.nan
        boxes = _unmold_boxes_for_test(rng, n, H, W)
        cls = rng.integers(0, C_, n).astype(np.int32)
        rows = rng.permutation(R)[:n].astype(np.int32)
        rows[:6] = (3, 4, 5, 3, 4, 5)
        boxes[:3] = ((0, 0, H, W), (10, 5, 60, 70), (2, 3, 50, 40))
        dets = np.concatenate([boxes, cls[:, None], rows[:, None]], axis=1).astype(np.int32)
        got = ops.unmold_masks(torch.tensor(mm, device=dev), torch.tensor(dets, device=dev), (H, W), packed=packed)
        torch.cuda.synchronize()
        got = got.cpu().numpy()
        if packed:
            assert got.shape == (H, W, 13)
            got = np.unpackbits(got, axis=-1, count=n, bitorder="little")
        assert got.shape == (H, W, n) and set(np.unique(got)) <= {0, 1}
        with np.errstate(invalid="ignore"):
            want = np.stack([utils.unmold_mask(mm[rows[i], :, :, cls[i]], boxes[i], (H, W, 3)) for i in range(n)], axis=-1)
        assert want[..., 0].all() and not want[..., 1].any() and not want[..., 2].any()      # the clip / NaN cases as intended
        np.testing.assert_array_equal(got.view(np.bool_), want)
    # no detections: an empty result, nothing launched
    e = ops.unmold_masks(torch.tensor(mm, device=dev), torch.zeros((0, 6), dtype=torch.int32, device=dev), (H, W), packed=packed)
    assert tuple(e.shape) == (H, W, 0)


def test_guarded_sgd_skips_non_finite_steps(dev):
    """Mixed-precision form of the optimiser step (mrcnn_sgd_momentum_guarded): bitwise the plain step while the squared
    gradient norm is finite; with an overflowed float16 gradient (inf / NaN in the buffer -> non-finite norm out of
    grad_prepare) weights and momentum stay bit-identical and the device counter moves."""
    ops = _ops()
    rng = np.random.default_rng(52)
    n = 64 * 40
    P, G, V = _rand(rng, n), _rand(rng, n, scale=3.0), _rand(rng, n, scale=0.1)
    gran = np.full(n // 64, 1e-5, np.float32)
    gd = torch.tensor(gran, device=dev)
    skipped = torch.zeros(1, dtype=torch.int32, device=dev)
    res = []
    for guarded in (False, True):
        Pd, Gd, Vd = (torch.tensor(a, device=dev) for a in (P, G, V))
        ss = torch.zeros(1, device=dev)
        ops.grad_prepare(Gd, Pd, 1.0, gd, ss)
        ops.sgd_momentum(Pd, Vd, Gd, ss, 5.0, 0.01, 0.9, gd, skipped=skipped if guarded else None)
        res.append((Pd.cpu().numpy(), Vd.cpu().numpy()))
    np.testing.assert_array_equal(res[0][0], res[1][0])
    np.testing.assert_array_equal(res[0][1], res[1][1])
    assert int(skipped.item()) == 0 and not np.array_equal(res[0][0], P)
    for k, poison in enumerate((np.inf, np.nan, -np.inf)):
        Gbad = G.copy(); Gbad[1234] = poison
        Pd, Gd, Vd = (torch.tensor(a, device=dev) for a in (P, Gbad, V))
        ss = torch.zeros(1, device=dev)
        ops.grad_prepare(Gd, Pd, 1.0, gd, ss)
        assert not np.isfinite(float(ss))
        ops.sgd_momentum(Pd, Vd, Gd, ss, 5.0, 0.01, 0.9, gd, skipped=skipped)
        np.testing.assert_array_equal(Pd.cpu().numpy(), P)
        np.testing.assert_array_equal(Vd.cpu().numpy(), V)
        assert int(skipped.item()) == k + 1


def test_sgd_step(dev):
    ops = _ops()
    rng = np.random.default_rng(51)
    sizes = [1000, 64, 4097, 3, 200]                 # the last tensor is frozen
    offs, off = [], 0
    for n in sizes:
        offs.append(off); off += (n + 63) // 64 * 64
    total = off
    P = np.zeros(total, np.float32); G = np.zeros(total, np.float32); V = np.zeros(total, np.float32)
    params, grads, vel = {}, {}, {}
    for i, (o, n) in enumerate(zip(offs, sizes)):
        params[i], grads[i], vel[i] = _rand(rng, n), _rand(rng, n, scale=3.0), _rand(rng, n, scale=0.1)
        P[o:o + n], G[o:o + n], V[o:o + n] = params[i], grads[i], vel[i]
    l2 = np.array([2e-4 / 1000, 0, 2e-4 / 4097, 0, -1], np.float32)
    gran = np.full(total // 64, -1, np.float32)
    for i, (o, n) in enumerate(zip(offs, sizes)):
        gran[o // 64:(o + n + 63) // 64] = l2[i]
    world = 2
    frozen_p, frozen_v = params.pop(4).copy(), vel.pop(4).copy()
    grads.pop(4)
    for i in params:
        grads[i] = grads[i] * np.float32(1.0 / world) + l2[i] * params[i]
    Pd, Gd, Vd, gd = (torch.tensor(a, device=dev) for a in (P, G, V, gran))
    ss = torch.zeros(1, device=dev); ss2 = torch.zeros(1, device=dev)
    ops.grad_prepare(Gd, Pd, 1.0 / world, gd, ss)
    ops.sumsq(Gd, ss2)
    ops.sgd_momentum(Pd, Vd, Gd, ss, 5.0, 0.01, 0.9, gd)
    torch.cuda.synchronize()
    norm = orc.sgd_step(params, grads, vel, 0.01, 0.9, 5.0)
    assert norm > 5.0
    np.testing.assert_allclose(float(ss.sqrt()), norm, rtol=1e-5)
    np.testing.assert_allclose(float(ss2), float(ss), rtol=1e-6)
    for i, (o, n) in enumerate(zip(offs, sizes)):
        if i == 4:
            assert np.array_equal(Pd[o:o + n].cpu().numpy(), frozen_p) and np.array_equal(Vd[o:o + n].cpu().numpy(), frozen_v)
            assert float(Gd[o:o + n].abs().max()) == 0.0
            continue
        np.testing.assert_allclose(Pd[o:o + n].cpu().numpy(), params[i], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(Vd[o:o + n].cpu().numpy(), vel[i], rtol=1e-5, atol=1e-6)


_FLAT_SCRIPT = r'''
import os, sys
import numpy as np, torch
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "oracle"))
import mrcnn_oracle as orc
import caesar_mrcnn_amd
from caesar_mrcnn_amd import ops
rng = np.random.default_rng(3)
dev = torch.device("cuda:0")
x = torch.tensor(rng.standard_normal((90, 13, 11, 128)).astype(np.float32), requires_grad=True)
w = torch.tensor((rng.standard_normal((3, 3, 128, 256)) / 34.0).astype(np.float32), requires_grad=True)
y = orc.conv2d_nhwc(x, w, None, 1, "same")
dy = torch.tensor(rng.standard_normal(tuple(y.shape)).astype(np.float32))
y.backward(dy)
got = ops.conv2d(x.detach().to(dev), w.detach().to(dev), None, None, None, None, 1, "same")
dw = ops.conv2d_wgrad(x.detach().to(dev), dy.to(dev), (3, 3, 128, 256), 1, "same")
torch.cuda.synchronize()
assert float((got.cpu() - y.detach()).abs().max()) <= 2e-4 * float(y.abs().max())
assert float((dw.cpu() - w.grad).abs().max()) <= 5e-4 * float(w.grad.abs().max())
print("flat-ok")
'''


def test_flat_addressed_lds_dma_kernels(dev):
    """conv_fwd_glds_kernel / conv_wgrad_glds_kernel serve tensors too large for a 32-bit buffer descriptor; the
    MRCNN_CONV_FLAT_GLDS switch (read once per process) routes an ordinary shape through them."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MRCNN_CONV_FLAT_GLDS="1")
    r = subprocess.run([sys.executable, "-c", _FLAT_SCRIPT % {"root": root}], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "flat-ok" in r.stdout, r.stderr[-2000:]


def test_conv_dgrad_fused_with_epilogue_backward_splitk(dev):
    """The same entry point on a small feature map: the split-K slab reduction carries the backward epilogue."""
    ops = _ops()
    rng = np.random.default_rng(91)
    # the last shape (98 x 2 tiles of 128 x 128, 144 K-steps of 16) takes the LDS-DMA kernel with split-K
    for (N, H, W, Cin, Cout, k) in ((4, 16, 16, 1024, 256, 1), (4, 16, 16, 256, 256, 3), (2, 8, 8, 512, 2048, 1), (64, 14, 14, 256, 256, 3)):
        dz = torch.tensor(_rand(rng, N, H, W, Cin), device=dev)
        wt = torch.tensor(_rand(rng, k, k, Cin, Cout, scale=1.0 / np.sqrt(k * k * Cin)), device=dev)
        below_out = torch.relu(torch.tensor(_rand(rng, N, H, W, Cout), device=dev))
        below_z = torch.tensor(_rand(rng, N, H, W, Cout), device=dev)
        scale, mean, rstd = (torch.tensor(rng.uniform(0.5, 1.5, Cout).astype(np.float32), device=dev) for _ in range(3))
        pad = ((k - 1) // 2, (k - 1) // 2) if k > 1 else "valid"
        sums = [torch.zeros(Cout, device=dev) for _ in range(3)]
        got = ops.conv2d_dgrad_ep(dz, wt, pad, below_out, below_z, scale, mean, rstd, sums[0], sums[1], sums[2], 1)
        assert got is not None, "expected the split-K path for %s" % ((N, H, W, Cin, Cout, k),)
        y = ops.conv2d(dz, wt, stride=1, padding=pad)
        ref = torch.empty_like(y)
        rs = [torch.zeros(Cout, device=dev) for _ in range(3)]
        ops.epilogue_bwd(y, below_out, below_z, scale, mean, rstd, None, ref, rs[0], rs[1], rs[2], 1)
        torch.cuda.synchronize()
        assert torch.equal(got, ref)
        for a, b in zip(sums, rs):
            torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-4 * float(b.abs().max()))


@pytest.mark.parametrize("k,res", [(3, False), (1, True)])
def test_conv_dgrad_fused_with_epilogue_backward(dev, k, res, tail_split_env):
    """mrcnn_conv2d_dgrad_ep == mrcnn_conv2d_fwd followed by mrcnn_epilogue_bwd (both checked against the oracle above):
    identical dz, channel sums equal up to the order of the float32 atomics; small layers report 'unsupported'."""
    ops = _ops()
    rng = np.random.default_rng(77 + k)
    N, H, W, Cin, Cout = 430, 14, 14, 128, 256                      # M = 84280 -> 659 x 2 tiles of 128 x 128
    dz = torch.tensor(_rand(rng, N, H, W, Cin), device=dev)
    wt = torch.tensor(_rand(rng, k, k, Cin, Cout, scale=1.0 / np.sqrt(k * k * Cin)), device=dev)
    below_out = torch.relu(torch.tensor(_rand(rng, N, H, W, Cout), device=dev))
    below_z = torch.tensor(_rand(rng, N, H, W, Cout), device=dev)
    r = torch.tensor(_rand(rng, N, H, W, Cout), device=dev) if res else None
    scale, mean, rstd = (torch.tensor(rng.uniform(0.5, 1.5, Cout).astype(np.float32), device=dev) for _ in range(3))
    pad = ((k - 1) // 2, (k - 1) // 2) if k > 1 else "valid"
    sums = [torch.zeros(Cout, device=dev) for _ in range(3)]
    got = ops.conv2d_dgrad_ep(dz, wt, pad, below_out, below_z, scale, mean, rstd, sums[0], sums[1], sums[2], 1, res=r)
    assert got is not None
    y = ops.conv2d(dz, wt, res=r, stride=1, padding=pad, res_mode=1 if res else 0)
    ref = torch.empty_like(y)
    rs = [torch.zeros(Cout, device=dev) for _ in range(3)]
    ops.epilogue_bwd(y, below_out, below_z, scale, mean, rstd, None, ref, rs[0], rs[1], rs[2], 1)
    torch.cuda.synchronize()
    assert torch.equal(got, ref)
    for a, b in zip(sums, rs):
        torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-4 * float(b.abs().max()))
    if k == 1:          # 4 images, K = 128: neither LDS-DMA tiles nor a split-K plan -> the caller must use the two calls
        small = ops.conv2d_dgrad_ep(dz[:4], wt, pad, below_out[:4], below_z[:4], scale, mean, rstd, sums[0], sums[1], sums[2], 1)
        assert small is None


def test_proposal_topk_preselection_fallback_on_massive_ties(dev):
    """A >= 32768 anchors go through the multi-workgroup pre-selection; when (almost) all scores are equal the
    candidates cannot be narrowed below the sorter's capacity and the single-workgroup selection takes over --
    top_k's tie rule (lowest anchor index first) must hold either way."""
    ops = _ops()
    rng = np.random.default_rng(123)
    A, limit, count = 40000, 6000, 50
    cfg = _cfg()
    ctr = rng.uniform(0, 1, (A, 2)); sz = np.exp(rng.uniform(np.log(0.02), np.log(0.5), (A, 2)))
    anchors = np.concatenate([ctr - sz / 2, ctr + sz / 2], 1).astype(np.float32)
    fg = np.full((2, A), 0.5, np.float32)
    fg[0, 100:300] = 0.9                                   # image 0: 200 distinct leaders, then 39 800 ties
    fg[1, :] = np.where(rng.uniform(0, 1, A) < 0.1, 0.75, 0.25)      # image 1: ~4000 at 0.75, the rest tie at 0.25
    probs = np.stack([1 - fg, fg], -1).astype(np.float32)
    deltas = _rand(rng, 2, A, 4, scale=0.1)
    _, top_idx, _, _, _ = ops.proposals(torch.tensor(probs, device=dev), torch.tensor(deltas, device=dev),
                                        torch.tensor(anchors, device=dev), limit, count, 0.7, cfg.RPN_BBOX_STD_DEV, debug=True)
    torch.cuda.synchronize()
    for b in range(2):
        ref = orc.tf_top_k_indices(fg[b], limit)
        assert np.array_equal(top_idx[b].cpu().numpy(), ref.astype(np.int32))


def _fits_tiles_for_test():
    """(name, float32 tile) cases for the device FITS path: the reference's two cut-outs (one with 288 NaN pixels), seeded
    synthetic tiles with a NaN border, +-inf pixels (-inf makes the NaN fill itself non-finite), fewer pixels than the 1000
    zscale samples, a sample count that is not a multiple of the stride, a constant tile and a heavy-tailed one (the rejection
    loop runs all its rounds)."""
    import os
    from caesar_mrcnn_amd import fits
    g = os.path.join(os.path.dirname(__file__), "golden")
    cases = [(n, fits.read_primary_hdu(os.path.join(g, n))[0].astype(np.float32)) for n in ("galaxy0002.fits", "sidelobe0001.fits")]
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:256, 0:256]
    t = rng.normal(0, 1, (256, 256)).astype(np.float32)
    for _ in range(5):
        cy, cx, sy, sx = rng.uniform(20, 236, 2).tolist() + rng.uniform(1.5, 12, 2).tolist()
        t += (np.exp(-0.5 * (((yy - cy) / sy) ** 2 + ((xx - cx) / sx) ** 2)) * rng.uniform(5, 200)).astype(np.float32)
    a = t.copy(); a[:3, :] = np.nan; a[:, -2:] = np.nan
    cases.append(("nan border", a))
    b = t.copy(); b[5, 7] = np.inf; b[100, 3] = np.inf; b[0, :40] = np.nan
    cases.append(("+inf pixels", b))
    c = t.copy(); c[9, 9] = -np.inf; c[200:203, :] = np.nan
    cases.append(("-inf pixel: the NaN fill is -inf", c))
    cases.append(("small 20 x 31", t[:20, :31].copy()))
    cases.append(("non-square 300 x 217", np.tile(t, (2, 1))[:300, :217].copy()))
    cases.append(("constant", np.full((64, 64), 3.5, np.float32)))
    cases.append(("heavy tail", (rng.standard_cauchy((128, 128)) * 3).astype(np.float32)))
    cases.append(("1024 x 1024", rng.normal(10, 2, (1024, 1024)).astype(np.float32)))
    return cases


def test_fits_to_rgb_device_equals_host(dev, tmp_path):
    """mrcnn_fits_to_rgb against the host statement of utils.read_fits (mrcnn/utils.py:1033-1163; caesar_mrcnn_amd.fits:
    NaN -> min, per-channel zscale, / max, round(255 x)): byte-identical uint8 images through fits.read_fits(device=...) --
    file parse and tile cut on the host, the big-endian floats swapped on the device -- for whole images and a cut tile,
    equal and unequal channel contrasts."""
    import warnings
    from caesar_mrcnn_amd import fits
    for k, (name, tile) in enumerate(_fits_tiles_for_test()):
        path = str(tmp_path / ("t%d.fits" % k))
        fits.write_fits(path, tile, {"BUNIT": "JY/BEAM"})
        for zc in ([0.25, 0.25, 0.25], [0.25, 0.3, 1.0]):
            with warnings.catch_warnings(), np.errstate(all="ignore"):
                warnings.simplefilter("ignore")
                want, _ = fits.read_fits(path, zscale_contrasts=zc)
                got, hdr = fits.read_fits(path, zscale_contrasts=zc, device=dev)
            assert got.dtype == np.uint8 and got.shape == tile.shape + (3,) and hdr["BUNIT"] == "JY/BEAM"
            diff = int((got != want).sum())
            assert diff == 0, "%s zc=%s: %d of %d bytes differ (max |d| %d)" % (
                name, zc, diff, got.size, int(np.abs(got.astype(int) - want.astype(int)).max()))
            if name != "constant":
                assert got.max() == 255 and len(np.unique(got)) > 8
        if tile.shape[0] >= 200:                                   # a tile cut out of the image (xmin / xmax / ymin / ymax)
            with warnings.catch_warnings(), np.errstate(all="ignore"):
                warnings.simplefilter("ignore")
                want, _ = fits.read_fits(path, xmin=17, xmax=149, ymin=30, ymax=162)
                got, _ = fits.read_fits(path, xmin=17, xmax=149, ymin=30, ymax=162, device=dev)
            assert got.shape == (132, 132, 3) and np.array_equal(got, want), name


def test_mold_inputs_device_equals_host(dev):
    """MaskRCNN._mold_inputs_device (mrcnn_mold_image_u8: bilinear up-scaling in float64, clip to the image's range, uint8
    truncation, zero padding, minus MEAN_PIXEL) against the host mold_inputs (mrcnn/model.py:2519-2556 -> utils.resize_image,
    mold_image): identical float32 canvas, metas and windows -- 132 -> 256 (the reference's cut-outs), non-square up-scaling,
    unscaled with padding, a grey (1-channel) image, a non-zero mean pixel; detect() on the device-molded input equals detect
    on the host-molded one."""
    import os
    from caesar_mrcnn_amd import fits
    from caesar_mrcnn_amd.config import run_py_config
    from caesar_mrcnn_amd.model import MaskRCNN
    rng = np.random.default_rng(9)
    cfg = run_py_config(backbone="custom", imgsize=256, mode="inference")
    cfg.POST_NMS_ROIS_INFERENCE = 200
    cfg.DETECTION_MAX_INSTANCES = 30
    model = MaskRCNN("inference", cfg, "/tmp/mrcnn_logs", device=dev, seed=5)
    cut, _ = fits.read_fits(os.path.join(os.path.dirname(__file__), "golden", "galaxy0002.fits"))
    images = [cut, rng.integers(0, 256, (100, 180, 3), dtype=np.uint8), rng.integers(0, 256, (256, 256, 3), dtype=np.uint8),
              rng.integers(0, 256, (256, 130, 3), dtype=np.uint8), rng.integers(20, 200, (77, 201, 3), dtype=np.uint8)]
    for mean in ([0.0, 0.0, 0.0], [123.7, 116.8, 103.9]):
        cfg.MEAN_PIXEL = np.array(mean)
        for img in images:
            want, wmeta, wwin = model.mold_inputs([img])
            got = model._mold_inputs_device([img])
            assert got is not None
            torch.cuda.synchronize()
            g = got[0].cpu().numpy()
            assert g.shape == want.shape and g.dtype == np.float32
            d = np.abs(g - want.astype(np.float32))
            assert d.max() == 0.0, (img.shape, mean, float(d.max()), int((d > 0).sum()))
            assert np.array_equal(got[1], wmeta) and np.array_equal(got[2], wwin)
    cfg.MEAN_PIXEL = np.array([0.0, 0.0, 0.0])
    assert model._mold_inputs_device([cut.astype(np.float32)]) is None            # not uint8: host path
    res_dev = model.detect([cut])[0]
    molded, metas, windows = model.mold_inputs([cut])
    out = model._run_graph(molded, metas)
    res_host = model._detect_results(out, [cut.shape], [molded[0].shape], windows)[0]
    for k in ("rois", "class_ids", "scores", "masks"):
        assert np.array_equal(res_dev[k], res_host[k]), k


def test_detect_batch_of_two_equals_single_images(dev):
    """detect() on a batch of two images of different original sizes (both molded to the 256 x 256 canvas on the device) returns,
    image by image, what two batch-1 calls return: same boxes, class ids, scores and full-size masks -- the per-slot pinned
    staging buffers, the per-image mrcnn_mask rows and the windows must not mix."""
    from caesar_mrcnn_amd.config import run_py_config
    from caesar_mrcnn_amd.model import MaskRCNN
    from caesar_mrcnn_amd.params import ParamLayout, init_weights
    rng = np.random.default_rng(41)
    imgs = [rng.integers(0, 256, (132, 132, 3), dtype=np.uint8), rng.integers(0, 256, (200, 256, 3), dtype=np.uint8)]
    for im in imgs:                                         # a few bright blobs so that the random network finds something
        for _ in range(4):
            y, x = rng.integers(10, im.shape[0] - 30), rng.integers(10, im.shape[1] - 30)
            im[y:y + 18, x:x + 22] = 250
    cfg1 = run_py_config(backbone="custom", imgsize=256, mode="inference")
    cfg1.POST_NMS_ROIS_INFERENCE = 200; cfg1.DETECTION_MAX_INSTANCES = 30
    w = init_weights(ParamLayout(cfg1), seed=13)
    m1 = MaskRCNN("inference", cfg1, "/tmp/mrcnn_logs", device=dev, weights=w)
    singles = [m1.detect([im])[0] for im in imgs]
    cfg2 = run_py_config(backbone="custom", imgsize=256, mode="inference")
    cfg2.IMAGES_PER_GPU = 2                                 # (run.py's InferenceConfig pins 1; BATCH_SIZE follows the attribute)
    cfg2.POST_NMS_ROIS_INFERENCE = 200; cfg2.DETECTION_MAX_INSTANCES = 30
    assert cfg2.BATCH_SIZE == 2
    m2 = MaskRCNN("inference", cfg2, "/tmp/mrcnn_logs", device=dev, weights=w)
    for rep in range(2):                                    # second call: replayed graph, reused staging buffers
        both = m2.detect(imgs)
        assert len(both) == 2
        for got, want, im in zip(both, singles, imgs):
            assert got["masks"].shape[:2] == im.shape[:2] and got["masks"].dtype == np.bool_
            for k in ("rois", "class_ids", "scores", "masks"):
                assert np.array_equal(got[k], want[k]), (rep, k)
    assert sum(r["rois"].shape[0] for r in singles) > 0
