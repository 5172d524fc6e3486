"""Host-side post-processing of detect results (scope row f3; mrcnn/analyze.py:1162-1496, 1866-2030).
Pure CPU: a stub stands in for MaskRCNN.detect.  skimage is absent here, so its restated helpers (contours, labelling)
are checked against their published semantics on hand-built cases (parity unpinned against the library); the clique
enumeration is pinned against networkx.find_cliques and the Jaccard index against sklearn.metrics.jaccard_score, both
importable in this image (skipped where they are not)."""
import json
import os

import numpy as np
import pytest

import caesar_mrcnn_amd  # noqa: F401
from caesar_mrcnn_amd import analyze
from caesar_mrcnn_amd.config import run_py_config


class _StubModel(object):
    def __init__(self, result):
        self.result = result

    def detect(self, images, verbose=0):
        return [self.result]


def _cfg():
    cfg = run_py_config(num_classes=4, imgsize=64, mode="inference")
    cfg.CLASS_NAMES = ["bkg", "sidelobe", "source", "galaxy"]
    return cfg


def _result(objs, shape=(64, 64)):
    n = len(objs)
    masks = np.zeros(shape + (n,), bool)
    rois = np.zeros((n, 4), np.int32)
    for i, (sl, _, _) in enumerate(objs):
        masks[sl + (i,)] = True
        ys, xs = np.nonzero(masks[:, :, i])
        rois[i] = (ys.min(), xs.min(), ys.max() + 1, xs.max() + 1)
    return {"rois": rois, "masks": masks, "class_ids": np.array([o[1] for o in objs], np.int32),
            "scores": np.array([o[2] for o in objs], np.float32)}


def test_label_components_raster_order_and_fallback(monkeypatch):
    m = np.zeros((6, 7), np.uint8)
    m[0, 5] = 1; m[1, 0:2] = 1; m[2, 1] = 1; m[4, 3:6] = 1; m[5, 6] = 1          # (5,6) touches (4,5) only diagonally
    lab, n = analyze.label_components(m)
    assert n == 4 and lab[0, 5] == 1 and lab[1, 0] == 2 and lab[2, 1] == 2 and lab[4, 4] == 3 and lab[5, 6] == 4
    import builtins
    real_import = builtins.__import__

    def no_scipy(name, *a, **k):
        if name.startswith("scipy"):
            raise ImportError(name)
        return real_import(name, *a, **k)
    monkeypatch.setattr(builtins, "__import__", no_scipy)
    lab2, n2 = analyze.label_components(m)
    assert n2 == n and np.array_equal(lab2, lab)


def test_find_contours_rectangle_and_hole():
    m = np.zeros((12, 14), np.uint8)
    m[3:8, 4:10] = 1                                                # 5 x 6 block
    cs = analyze.find_contours(m, 0.5)
    assert len(cs) == 1
    c = cs[0]
    assert np.array_equal(c[0], c[-1]) and len(c) == 2 * (5 + 6) + 1
    assert np.all((c * 2) == np.round(c * 2))                       # binary image: every vertex on a half-pixel
    assert c[:, 0].min() == 2.5 and c[:, 0].max() == 7.5 and c[:, 1].min() == 3.5 and c[:, 1].max() == 9.5
    y, x = c[:, 0], c[:, 1]
    area = 0.5 * np.sum(x[:-1] * y[1:] - x[1:] * y[:-1])
    assert abs(abs(area) - (5 * 6 - 4 * 0.125)) < 1e-9             # corners are cut at 45 degrees
    m[5, 6] = 0                                                     # a hole -> second contour, opposite orientation
    cs = analyze.find_contours(m, 0.5)
    assert len(cs) == 2
    areas = [0.5 * np.sum(c[:-1, 1] * c[1:, 0] - c[1:, 1] * c[:-1, 0]) for c in cs]
    assert areas[0] * areas[1] < 0 and min(abs(a) for a in areas) == 0.5
    # interpolation on a non-binary field
    f = np.array([[0., 0., 0.], [0., 2., 0.], [0., 0., 0.]])
    c = analyze.find_contours(f, 0.5)[0]
    assert np.allclose(sorted(c[:-1].tolist()), sorted([[0.25, 1.0], [1.0, 0.25], [1.75, 1.0], [1.0, 1.75]]))


def test_cliques_components_iou():
    adj = {0: {1, 2}, 1: {0, 2}, 2: {0, 1, 3}, 3: {2}}
    assert sorted(analyze.maximal_cliques(adj)) == [[0, 1, 2], [2, 3]]
    assert analyze.maximal_cliques({}) == []
    assert analyze.connected_components(5, [(0, 3), (3, 1)]) == [[0, 3, 1], [2], [4]]
    a = np.zeros((4, 4), bool); b = np.zeros((4, 4), bool)
    a[:2] = True; b[1:3] = True
    assert analyze.mask_iou(a, b) == pytest.approx(4 / 12) and analyze.mask_iou(a * 0, b * 0) == 0.0


def test_extract_det_masks_merge_select_and_outputs(tmp_path):
    cfg = _cfg()
    objs = [
        ((slice(5, 15), slice(5, 15)), 2, 0.95),     # 0 source A
        ((slice(8, 18), slice(5, 15)), 2, 0.85),     # 1 source overlapping A with IoU 70/130 -> merged with 0
        ((slice(30, 40), slice(30, 40)), 2, 0.90),   # 2 source B
        ((slice(32, 42), slice(32, 42)), 3, 0.99),   # 3 galaxy overlapping B (other class) -> best score wins
        ((slice(50, 55), slice(50, 55)), 1, 0.60),   # 4 below the score threshold
        ((slice(0, 6), slice(56, 64)), 1, 0.80),     # 5 isolated sidelobe touching the border
        ((slice(20, 24), slice(5, 9)), 2, 0.75),     # 6 source, 4-connected to nothing
    ]
    an = analyze.Analyzer(_StubModel(_result(objs)), cfg)
    an.outfile_json = str(tmp_path / "o.json")
    an.outfile_ds9 = str(tmp_path / "o.reg")
    an.obj_name_tag = "t7"
    image = np.zeros((64, 64, 3), np.uint8)
    assert an.predict(image, image_id="img1", xmin=100, ymin=200) == 0
    # order: score-sorted selection (3, 0, 2, 1, 5, 6) -> merge (0+1) -> cliques {2,3} keep 3
    names = an.class_names_final
    assert names == ["galaxy", "source", "sidelobe", "source"]
    assert an.scores_final[0] == pytest.approx(0.99) and an.scores_final[1] == pytest.approx(0.90)   # mean(0.95, 0.85)
    merged = an.masks_final[1]
    assert merged.sum() == 130 and np.array_equal(an.bboxes[1], [5, 5, 18, 15])
    assert an.captions[1] == "source 0.90"
    res = json.load(open(an.outfile_json))
    assert res["image_id"] == "img1" and len(res["objs"]) == 4
    o = res["objs"][1]
    assert set(o) == {"name", "x1", "x2", "y1", "y2", "class_id", "class_name", "score", "pixels", "vertexes", "edge"}
    assert o["name"] == "S2_t7" and (o["x1"], o["x2"], o["y1"], o["y2"]) == (105, 115, 205, 218) and o["class_id"] == 2
    assert len(o["pixels"]) == 130 and o["pixels"][0] == [205, 105] and o["edge"] is False
    assert len(o["vertexes"]) == 1 and o["vertexes"][0][0] == o["vertexes"][0][-1]
    xs = [v[0] for v in o["vertexes"][0]]; ys = [v[1] for v in o["vertexes"][0]]
    assert (min(xs), max(xs), min(ys), max(ys)) == (104.5, 114.5, 204.5, 217.5)
    assert res["objs"][2]["edge"] is True and res["objs"][2]["class_name"] == "sidelobe"
    reg = open(an.outfile_ds9).read().splitlines()
    assert reg[2] == "image" and len(reg) == 3 + 4 and reg[4].startswith("polygon(") and "tag={source}" in reg[4]
    assert "tag={BORDER}" in reg[5] and "color=red" in reg[5]


def test_split_masks_and_no_detection(tmp_path):
    cfg = _cfg()
    two_blobs = np.zeros((64, 64), bool)
    two_blobs[5:10, 5:10] = True
    two_blobs[30:36, 30:36] = True
    res = {"rois": np.array([[5, 5, 36, 36], [5, 5, 36, 36]], np.int32), "masks": np.stack([two_blobs, np.roll(two_blobs, 12, 1)], -1),
           "class_ids": np.array([2, 3], np.int32), "scores": np.array([0.9, 0.8], np.float32)}
    an = analyze.Analyzer(_StubModel(res), cfg)
    an.split_masks = True
    an.write_to_json = an.write_to_ds9 = False
    assert an.predict(np.zeros((64, 64, 3), np.uint8), image_id="x") == 0
    assert an.class_names_final == ["source", "source", "galaxy"]              # the galaxy mask is never split
    assert [int(np.asarray(m).sum()) for m in an.masks_final] == [25, 36, 61]
    empty = {"rois": np.zeros((0, 4), np.int32), "masks": np.zeros((64, 64, 0), bool), "class_ids": np.zeros(0, np.int32),
             "scores": np.zeros(0, np.float32)}
    an = analyze.Analyzer(_StubModel(empty), cfg)
    assert an.predict(np.zeros((64, 64, 3), np.uint8), image_id="y") == 0 and an.masks_final == [] and an.results == {}
    assert not os.path.exists("out_y.json")


def test_maximal_cliques_match_networkx():
    """analyze.py:1332-1362 builds an nx.Graph with add_edge and takes list(nx.find_cliques(g)); the clique ORDER is
    irrelevant downstream (cliques are re-sorted by their best score, analyze.py:1378), so sets of sets are compared."""
    nx = pytest.importorskip("networkx")
    rng = np.random.default_rng(5)
    for n, p in ((6, 0.5), (12, 0.3), (25, 0.2), (40, 0.12), (9, 1.0)):
        edges = [(i, j) for i in range(n) for j in range(i + 1, n) if rng.uniform() < p]
        g = nx.Graph()
        adj = {}
        for v, w in edges:
            g.add_edge(v, w)
            adj.setdefault(v, set()).add(w)
            adj.setdefault(w, set()).add(v)
        ref = {frozenset(c) for c in nx.find_cliques(g)}
        got = analyze.maximal_cliques(adj)
        assert len(got) == len(ref) and {frozenset(c) for c in got} == ref


def test_mask_iou_matches_sklearn_jaccard():
    """analyze.py:29 imports sklearn.metrics.jaccard_score for mask overlaps (average='binary' on the flattened boolean masks, analyze.py:1271)."""
    skm = pytest.importorskip("sklearn.metrics")
    rng = np.random.default_rng(9)
    for _ in range(5):
        a = rng.uniform(size=(24, 24)) < 0.4
        b = rng.uniform(size=(24, 24)) < 0.5
        ref = skm.jaccard_score(a.ravel(), b.ravel())
        assert abs(analyze.mask_iou(a, b) - ref) < 1e-12


def test_find_contours_vertices_match_contourpy():
    """Second source for the marching-squares restatement (skimage.measure.find_contours itself is not installable here):
    contourpy (matplotlib's contouring engine, importable in the build image) must produce exactly the same iso-line vertices
    -- on random fields (every square type incl. saddles: only the way saddle segments are JOINED is a convention, the
    vertices are not) -- and, on a smooth saddle-free field, the same closed polylines (same vertex set per line, same
    enclosed area)."""
    contourpy = pytest.importorskip("contourpy")
    rng = np.random.default_rng(11)

    def vertex_set(lines, swap):
        return sorted(set((round(float(p[1 if swap else 0]), 9), round(float(p[0 if swap else 1]), 9)) for l in lines for p in l))
    for shape in ((12, 15), (28, 28), (7, 40)):
        f = rng.random(shape)
        for level in (0.5, 0.31):
            theirs = contourpy.contour_generator(z=f, corner_mask=False).lines(level)      # (x, y) = (col, row)
            mine = analyze.find_contours(f, level)
            assert vertex_set(theirs, True) == vertex_set(mine, False), (shape, level)
    yy, xx = np.mgrid[0:40, 0:48]
    g = np.exp(-0.5 * (((yy - 12) / 4.0) ** 2 + ((xx - 14) / 6.0) ** 2)) + 0.8 * np.exp(-0.5 * (((yy - 28) / 5.0) ** 2 + ((xx - 33) / 3.5) ** 2))
    theirs = contourpy.contour_generator(z=g, corner_mask=False).lines(0.5)
    mine = analyze.find_contours(g, 0.5)
    assert len(theirs) == len(mine) == 2
    area = lambda y, x: abs(0.5 * np.sum(x[:-1] * y[1:] - x[1:] * y[:-1]))
    a = sorted((round(area(l[:, 1], l[:, 0]), 9), len(l)) for l in theirs)
    b = sorted((round(area(c[:, 0], c[:, 1]), 9), len(c)) for c in mine)
    assert a == b
    bm = (g > 0.5).astype(np.uint8)                                   # the binary masks Analyzer actually contours
    assert vertex_set(contourpy.contour_generator(z=bm.astype(float), corner_mask=False).lines(0.5), True) == \
        vertex_set(analyze.find_contours(bm, 0.5), False)
