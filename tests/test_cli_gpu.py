"""End-to-end through the command line with the reference's flags: synthetic FITS dataset ->
`run.py train` (writes a Keras-HDF5 checkpoint) -> `run.py test` and `run.py detect` loading it."""
import glob
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make_dataset(d, n=4, size=128):
    sys.path.insert(0, ROOT)
    from caesar_mrcnn_amd import fits
    rng = np.random.RandomState(0)
    rows = []
    yy, xx = np.mgrid[0:size, 0:size]
    for i in range(n):
        img = rng.normal(0, 1, (size, size)).astype(np.float32)
        cy, cx = rng.uniform(30, size - 30, 2)
        blob = np.exp(-0.5 * (((yy - cy) / 6.0) ** 2 + ((xx - cx) / 9.0) ** 2))
        img += 80 * blob
        if i == 0:
            img[:3, :] = np.nan
        fits.write_fits(os.path.join(d, "img%d.fits" % i), img, {"BUNIT": "JY/BEAM"})
        fits.write_fits(os.path.join(d, "mask%d.fits" % i), (blob > 0.3).astype(np.float32))
        rows.append("%s,%s,%s" % (os.path.join(d, "img%d.fits" % i), os.path.join(d, "mask%d.fits" % i),
                                  ["source", "galaxy"][i % 2]))
    lst = os.path.join(d, "train.dat")
    open(lst, "w").write("\n".join(rows) + "\n")
    return lst


def _run(args, cwd):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "run.py")] + args, cwd=cwd, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=280)
    out = r.stdout.decode()
    assert r.returncode == 0, out[-3000:]
    return out


def test_train_test_detect_cli(dev, tmp_path):
    d = str(tmp_path)
    lst = _make_dataset(d)
    common = ["--imgsize", "128", "--backbone", "custom", "--logs", os.path.join(d, "logs"), "--datalist", lst,
              "--train_rois_per_image", "32", "--rpn_train_anchors_per_image", "64", "--max_gt_instances", "10"]
    out = _run(["train"] + common + ["--nepochs", "1", "--epoch_length", "2", "--nvalidation_steps", "1",
                                     "--nimg_per_gpu", "2"], d)
    assert "Epoch 1/1" in out
    ckpts = glob.glob(os.path.join(d, "logs", "rg-dataset*", "mask_rcnn_rg-dataset_0001.h5"))
    assert len(ckpts) == 1 and open(ckpts[0], "rb").read(4) == b"\x89HDF"
    out = _run(["test"] + common + ["--weights", ckpts[0], "--scoreThr", "0.0"], d)
    assert "completeness" in out and "class galaxy" in out
    outjson = os.path.join(d, "det.json")
    out = _run(["detect"] + common + ["--weights", ckpts[0], "--image", os.path.join(d, "img0.fits"), "--scoreThr", "0.0",
                                      "--detect_outfile_json", outjson], d)
    res = json.load(open(outjson))
    assert res["image_id"] == "img0" and isinstance(res["objs"], list)        # the reference's out_<id>.json layout
    for s in res["objs"][:3]:
        assert {"name", "x1", "x2", "y1", "y2", "class_id", "class_name", "score", "pixels", "vertexes", "edge"} == set(s)
    if res["objs"]:
        assert open(os.path.join(d, "det.reg")).read().splitlines()[2] == "image"
    # resume naming: the checkpoint path carries the epoch the reference parses back (model.py:2375-2383)
    assert "Re-starting from epoch 1" in out
