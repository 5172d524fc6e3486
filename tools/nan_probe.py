import os, sys
sys.path.insert(0, "/root/repo")
import torch, bench
from caesar_mrcnn_amd import ops
from caesar_mrcnn_amd.config import run_py_config
from caesar_mrcnn_amd.model import MaskRCNN
dev = torch.device("cuda", 0)
cfg = run_py_config(num_classes=4, imgsize=512, backbone="resnet101", images_per_gpu=4, gpu_count=1)
model = MaskRCNN("training", cfg, "/tmp/mrcnn_bench_logs", device=dev, seed=0)
model.compile(cfg.LEARNING_RATE, cfg.LEARNING_MOMENTUM)
inp = model._to_device(bench.synthetic_batch(cfg, 4, seed=1234))
eng = model.engine
eng.sparse_mask_bwd = False
eng.head_dtype = torch.float16
watch = False
hits = 0
def fin(t): return t is None or bool(torch.isfinite(t.float()).all())
def amax(t): return float(t.float().abs().max()) if t is not None else 0.0
def wrap(name):
    orig = getattr(ops, name)
    def f(*a, **k):
        r = orig(*a, **k)
        if watch:
            torch.cuda.synchronize()
            ins = [x for x in a if torch.is_tensor(x)]
            outs = [x for x in (r if isinstance(r, (tuple, list)) else [r]) if torch.is_tensor(x)]
            if not all(fin(x) for x in ins + outs):
                global hits; hits += 1
                print("  %s: inputs %s  outputs %s" % (name, [(tuple(x.shape), x.dtype, fin(x), "%.3g" % amax(x)) for x in ins],
                                                     [(tuple(x.shape), fin(x)) for x in outs]), flush=True)
        return r
    setattr(ops, name, f)
for n in ("conv2d_wgrad_h16", "axpy_from_h16", "mask_out_bwd_h16", "epilogue_bwd_h16", "conv2d_h16", "cast_from_h16", "cast_to_h16", "mask_out_fwd_h16", "roialign"):
    if hasattr(ops, n): wrap(n)
for step in range(16):
    watch = True
    if watch: print("step", step)
    eng.forward_backward(*inp); eng.apply_gradients(cfg.LEARNING_RATE, cfg.LEARNING_MOMENTUM, 1)
    torch.cuda.synchronize()
    if hits > 6: break
    if watch: print(" params finite:", fin(eng.params), " grads finite:", fin(eng.grads) if hasattr(eng, "grads") else "?")
