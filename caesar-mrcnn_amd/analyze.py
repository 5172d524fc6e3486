"""Post-processing of ``MaskRCNN.detect`` results into the user-visible artefacts (row f3 of the scope
table): score filter, optional splitting of non-connected masks, merging of connected same-class masks,
selection of the best mask among overlapping ones, JSON and DS9-region output.  Restates
``Analyzer.predict / extract_det_masks / make_json_results / make_ds9_regions`` of mrcnn/analyze.py
(:833-905, :1162-1496, :1866-2030) and mrcnn/graph.py.  Host-side NumPy; the only device work is the
``model.detect`` call.

Third-party pieces the reference imports and this file restates (absent here, so parity is UNPINNED for
them; the semantics are the published ones):
  * skimage.measure.label(connectivity=1)      -> ``label_components`` (4-connected, raster-order numbering)
  * skimage.measure.find_contours(level=0.5)   -> ``find_contours`` (marching squares, 'low' connectivity)
  * sklearn.metrics.jaccard_score(binary)      -> ``mask_iou``
  * networkx.find_cliques                      -> ``maximal_cliques`` (Bron-Kerbosch with pivoting)
  * regions (DS9 writer)                       -> plain-text DS9 ``polygon`` / ``box`` lines, image coordinates
"""
import json
import logging
from collections import deque

import numpy as np

from . import utils

logger = logging.getLogger("mrcnn")

NO_SPLIT_CLASSES = ('galaxy_C2', 'galaxy_C3', 'galaxy', 'extended-multisland')      # analyze.py:1221


# ---- restated third-party helpers -----------------------------------------------------------------------
def label_components(mask):
    """4-connected components of a binary mask, numbered 1..n in raster order of their first pixel."""
    m = np.asarray(mask) != 0
    try:
        from scipy import ndimage
        lab, n = ndimage.label(m)
        return lab, int(n)
    except ImportError:
        lab = np.zeros(m.shape, np.int32)
        n = 0
        H, W = m.shape
        for r, c in zip(*np.nonzero(m)):
            if lab[r, c]:
                continue
            n += 1
            lab[r, c] = n
            q = deque([(r, c)])
            while q:
                y, x = q.popleft()
                for yy, xx in ((y - 1, x), (y + 1, x), (y, x - 1), (y, x + 1)):
                    if 0 <= yy < H and 0 <= xx < W and m[yy, xx] and not lab[yy, xx]:
                        lab[yy, xx] = n
                        q.append((yy, xx))
        return lab, n


def mask_iou(a, b):
    a, b = np.asarray(a).astype(bool).ravel(), np.asarray(b).astype(bool).ravel()
    union = np.logical_or(a, b).sum()
    return float(np.logical_and(a, b).sum()) / float(union) if union else 0.0


def find_contours(arr, level=0.5):
    """Iso-contours of a 2-D array by marching squares: list of [n, 2] (row, col) arrays, closed contours
    repeat their first point.  Squares are visited in raster order; segments keep the higher values on
    their left; ambiguous saddles connect the low corners (skimage default ``fully_connected='low'``)."""
    a = np.asarray(arr, dtype=np.float64)
    H, W = a.shape
    segs = []

    def frac(v0, v1):
        return (level - v0) / (v1 - v0)

    hi = a > level
    case = (hi[:-1, :-1] * 1 + hi[:-1, 1:] * 2 + hi[1:, :-1] * 4 + hi[1:, 1:] * 8)
    for r, c in zip(*np.nonzero((case != 0) & (case != 15))):
        ul, ur, ll, lr = a[r, c], a[r, c + 1], a[r + 1, c], a[r + 1, c + 1]
        k = case[r, c]
        top = (float(r), c + frac(ul, ur)) if (k & 1) != ((k >> 1) & 1) else None
        bottom = (r + 1.0, c + frac(ll, lr)) if ((k >> 2) & 1) != ((k >> 3) & 1) else None
        left = (r + frac(ul, ll), float(c)) if (k & 1) != ((k >> 2) & 1) else None
        right = (r + frac(ur, lr), c + 1.0) if ((k >> 1) & 1) != ((k >> 3) & 1) else None
        table = {1: [(top, left)], 2: [(right, top)], 3: [(right, left)], 4: [(left, bottom)], 5: [(top, bottom)],
                 6: [(right, top), (left, bottom)], 7: [(right, bottom)], 8: [(bottom, right)],
                 9: [(top, left), (bottom, right)], 10: [(bottom, top)], 11: [(bottom, left)], 12: [(left, right)],
                 13: [(top, right)], 14: [(left, top)]}
        segs.extend(table[int(k)])

    contours, starts, ends, nxt = {}, {}, {}, 0
    for p, q in segs:
        if p == q:
            continue
        tail, tail_num = starts.pop(q, (None, None))
        head, head_num = ends.pop(p, (None, None))
        if tail is not None and head is not None:
            if tail is head:
                head.append(q)                                  # closes the contour
            elif tail_num > head_num:
                head.extend(tail)
                contours.pop(tail_num, None)
                starts[head[0]] = (head, head_num)
                ends[head[-1]] = (head, head_num)
            else:
                tail.extendleft(reversed(head))
                contours.pop(head_num, None)
                starts[tail[0]] = (tail, tail_num)
                ends[tail[-1]] = (tail, tail_num)
        elif tail is None and head is None:
            new = deque((p, q))
            contours[nxt] = new
            starts[p] = (new, nxt)
            ends[q] = (new, nxt)
            nxt += 1
        elif head is None:
            tail.appendleft(p)
            starts[p] = (tail, tail_num)
        else:
            head.append(q)
            ends[q] = (head, head_num)
    return [np.array(c) for _, c in sorted(contours.items())]


def maximal_cliques(adj):
    """All maximal cliques of an undirected graph {node: set(neighbours)} (nodes without edges are absent,
    as in a networkx graph built with add_edge only)."""
    out = []

    def expand(R, P, X):
        if not P and not X:
            out.append(sorted(R))
            return
        pivot = max(P | X, key=lambda u: len(adj[u] & P))
        for v in sorted(P - adj[pivot]):
            expand(R | {v}, P & adj[v], X & adj[v])
            P = P - {v}
            X = X | {v}

    if adj:
        expand(set(), set(adj), set())
    return out


def connected_components(n, edges):
    """mrcnn/graph.py: depth-first components in vertex order, neighbours in insertion order."""
    adj = [[] for _ in range(n)]
    for v, w in edges:
        adj[v].append(w)
        adj[w].append(v)
    seen, cc = [False] * n, []
    for v in range(n):
        if seen[v]:
            continue
        comp = []
        # recursive DFS order of the reference: visit v, then each unvisited neighbour in turn
        def dfs(u):
            seen[u] = True
            comp.append(u)
            for w in adj[u]:
                if not seen[w]:
                    dfs(w)
        dfs(v)
        cc.append(comp)
    return cc


# ---- the Analyzer ------------------------------------------------------------------------------------------
class Analyzer(object):
    """Same attribute names and defaults as the reference's Analyzer (analyze.py:560-712) for the
    ``predict`` path."""

    def __init__(self, model, config, dataset=None):
        self.model = model
        self.config = config
        self.dataset = dataset
        self.n_classes = config.NUM_CLASSES
        self.class_names = config.CLASS_NAMES
        self.image = None
        self.image_id = -1
        self.image_header = None
        self.image_xmin = 0
        self.image_ymin = 0
        self.masks = self.boxes = self.class_ids = self.scores = None
        self.nobjects = 0
        self.masks_final, self.class_ids_final, self.class_names_final, self.scores_final = [], [], [], []
        self.bboxes, self.captions, self.bboxes_gt = [], [], []
        self.split_masks = False
        self.merge_overlapped_masks = True
        self.select_best_overlapped_masks = True
        self.split_source_sidelobe = True
        self.merge_overlap_iou_thr = 0.3
        self.results = {}
        self.obj_name_tag = ""
        self.obj_regions = []
        self.score_thr = 0.7
        self.iou_thr = 0.6
        self.outfile = self.outfile_json = self.outfile_ds9 = ""
        self.draw = False                    # plotting is out of scope (no matplotlib on the path)
        self.write_to_json = True
        self.write_to_ds9 = True
        self.use_polygon_regions = True
        self.class_color_map_ds9 = {'bkg': "black", 'spurious': "red", 'compact': "blue", 'extended': "green",
                                    'extended-multisland': "orange", 'flagged': "magenta",
                                    'sidelobe': "red", 'source': "blue", 'galaxy': "yellow"}

    # ---- mask helpers (analyze.py:2142-2171) ---------------------------------------------------------------
    def merge_masks(self, mask1, mask2):
        mask = mask1 + mask2
        mask[mask > 1] = 1
        return mask

    def extract_mask_connected_components(self, mask):
        return label_components(mask)

    def are_mask_connected(self, mask1, mask2):
        _, n1 = label_components(mask1)
        _, n2 = label_components(mask2)
        _, n = label_components(self.merge_masks(mask1, mask2))
        return n != n1 + n2

    # ---- predict (analyze.py:833-905) ----------------------------------------------------------------------
    def predict(self, image, image_id='', bboxes_gt=[], header=None, xmin=0, ymin=0):
        if image is None:
            logger.error("No input image given!")
            return -1
        self.image, self.image_xmin, self.image_ymin = image, xmin, ymin
        if image_id:
            self.image_id = image_id
        if header:
            self.image_header = header
        r = self.model.detect([self.image], verbose=0)[0]
        self.class_names = self.config.CLASS_NAMES
        self.masks, self.boxes, self.class_ids, self.scores = r['masks'], r['rois'], r['class_ids'], r['scores']
        self.nobjects = self.masks.shape[-1]
        self.results, self.obj_regions = {}, []
        if self.nobjects > 0:
            self.extract_det_masks()
        else:
            self.masks_final, self.class_ids_final, self.class_names_final, self.scores_final = [], [], [], []
            self.bboxes, self.captions = [], []
            logger.warning("No detected object found for image %s ..." % self.image_id)
            return 0
        self.bboxes_gt = bboxes_gt
        self.make_json_results()
        if self.write_to_json:
            self.write_json_results(self.outfile_json or 'out_' + str(self.image_id) + '.json')
        self.make_ds9_regions(self.use_polygon_regions)
        if self.write_to_ds9:
            self.write_ds9_regions(self.outfile_ds9 or 'out_' + str(self.image_id) + '.reg')
        return 0

    # ---- extract_det_masks (analyze.py:1162-1496) ----------------------------------------------------------
    def extract_det_masks(self):
        self.masks_final, self.class_ids_final, self.class_names_final, self.scores_final = [], [], [], []
        self.bboxes, self.captions = [], []
        masks_sel, class_ids_sel, scores_sel = [], [], []
        for i in range(self.boxes.shape[0]):
            if self.scores[i] < self.score_thr:
                continue
            masks_sel.append(self.masks[:, :, i])
            class_ids_sel.append(self.class_ids[i])
            scores_sel.append(self.scores[i])
        order = np.argsort(scores_sel)[::-1]

        masks_det, class_ids_det, scores_det = [], [], []
        for index in order:
            mask, class_id, score = masks_sel[index], class_ids_sel[index], scores_sel[index]
            if not self.split_masks or self.class_names[class_id] in NO_SPLIT_CLASSES:
                masks_det.append(mask); class_ids_det.append(class_id); scores_det.append(score)
                continue
            labels, ncomp = label_components(mask)
            for k in range(ncomp):
                masks_det.append(np.where(labels == k + 1, [1], [0]))
                class_ids_det.append(class_id); scores_det.append(score)

        masks_merged, class_ids_merged, scores_merged = [], [], []
        if self.merge_overlapped_masks:
            N = len(masks_det)
            edges = []
            for i in range(N):
                for j in range(i + 1, N):
                    if class_ids_det[i] != class_ids_det[j]:
                        continue
                    if mask_iou(masks_det[i], masks_det[j]) < self.merge_overlap_iou_thr:
                        continue
                    if self.are_mask_connected(masks_det[i], masks_det[j]):
                        edges.append((i, j))
            for comp in connected_components(N, edges):
                merged = masks_det[comp[0]]
                for index in comp[1:]:
                    merged = self.merge_masks(merged, masks_det[index])
                masks_merged.append(merged)
                class_ids_merged.append(class_ids_det[comp[-1]])
                score_avg = 0
                for index in comp:
                    score_avg += scores_det[index]
                scores_merged.append(score_avg * (1. / len(comp)))
        else:
            masks_merged, class_ids_merged, scores_merged = list(masks_det), list(class_ids_det), list(scores_det)

        is_selected = [True] * len(masks_merged)
        if self.select_best_overlapped_masks:
            adj = {}
            n_final = len(masks_merged)
            for i in range(n_final):
                label_i = self.class_names[class_ids_merged[i]]
                for j in range(i + 1, n_final):
                    label_j = self.class_names[class_ids_merged[j]]
                    if not self.are_mask_connected(masks_merged[i], masks_merged[j]):
                        continue
                    sidelobe_other = (label_i == 'spurious') != (label_j == 'spurious')
                    if self.split_source_sidelobe and sidelobe_other and \
                            mask_iou(masks_merged[i], masks_merged[j]) < self.merge_overlap_iou_thr:
                        continue
                    adj.setdefault(i, set()).add(j)
                    adj.setdefault(j, set()).add(i)
            cliques = maximal_cliques(adj)
            best = []
            for clique in cliques:
                max_score, max_index = -1, -1
                for index in clique:
                    if scores_merged[index] > max_score:
                        max_score, max_index = scores_merged[index], index
                best.append((max_score, max_index))
            for k in sorted(range(len(cliques)), key=lambda k: best[k][0], reverse=True):
                for index in cliques[k]:
                    if index != best[k][1] and is_selected[index]:
                        is_selected[index] = False

        for index in range(len(masks_merged)):
            if not is_selected[index]:
                continue
            m = masks_merged[index]
            bbox = utils.extract_bboxes(np.asarray(m).astype(bool)[:, :, None])
            if bbox[0][1] >= bbox[0][3] or bbox[0][0] >= bbox[0][2]:
                logger.warning("Invalid det bbox(%d,%d,%d,%d), skip it ..." % (bbox[0][1], bbox[0][3], bbox[0][0], bbox[0][2]))
                continue
            label = self.class_names[class_ids_merged[index]]
            self.masks_final.append(m)
            self.class_ids_final.append(class_ids_merged[index])
            self.class_names_final.append(label)
            self.scores_final.append(scores_merged[index])
            self.bboxes.append(bbox[0])
            self.captions.append("{} {:.2f}".format(label, scores_merged[index]))

    # ---- JSON (analyze.py:1866-1955) -----------------------------------------------------------------------
    def make_json_results(self):
        self.results = {"image_id": self.image_id, "objs": []}
        xmin, ymin = self.image_xmin, self.image_ymin
        ny, nx = self.image.shape[0], self.image.shape[1]
        for i in range(len(self.masks_final)):
            class_id = int(self.class_ids_final[i])
            y1, x1, y2, x2 = [int(v) for v in self.bboxes[i]]
            at_edge = (x1 <= 0 or x1 >= nx - 1 or x2 <= 0 or x2 >= nx - 1 or
                       y1 <= 0 or y1 >= ny - 1 or y2 <= 0 or y2 >= ny - 1)
            mask = np.asarray(self.masks_final[i])
            pixels = (np.argwhere(mask == 1) + np.array([ymin, xmin])).tolist()
            padded = np.zeros((mask.shape[0] + 2, mask.shape[1] + 2), dtype=np.uint8)
            padded[1:-1, 1:-1] = mask
            vertexes = []
            for verts in find_contours(padded, 0.5):
                verts = np.fliplr(verts) - 1                      # drop the padding, (y, x) -> (x, y)
                vertexes.append((verts + np.array([xmin, ymin])).tolist())
            self.results["objs"].append({
                "name": 'S' + str(i + 1) + "_" + self.obj_name_tag,
                "x1": xmin + x1, "x2": xmin + x2, "y1": ymin + y1, "y2": ymin + y2,
                "class_id": class_id, "class_name": self.class_names[class_id],
                "score": float(self.scores_final[i]), "pixels": pixels, "vertexes": vertexes, "edge": bool(at_edge)})

    def write_json_results(self, outfile):
        if not self.results:
            logger.warning("Result obj dictionary is empty, nothing to be written...")
            return
        with open(outfile, 'w') as fp:
            json.dump(self.results, fp, indent=2, sort_keys=True)

    # ---- DS9 regions (analyze.py:1960-2030) ----------------------------------------------------------------
    def make_ds9_regions(self, use_polygon=True):
        """One region line per contour (or the bounding box); DS9 image coordinates are 1-based."""
        self.obj_regions = []
        if not self.results or 'objs' not in self.results:
            logger.warning("No result dictionary was filled or no object detected, no region will be produced...")
            return -1
        for o in self.results['objs']:
            dx, dy = o['x2'] - o['x1'], o['y2'] - o['y1']
            xc, yc = o['x1'] + 0.5 * dx, o['y1'] + 0.5 * dy
            tags = "tag={%s}" % o['class_name'] + (" tag={BORDER}" if o['edge'] else "")
            meta = "# text={%s} %s color=%s" % (o['name'], tags, self.class_color_map_ds9.get(o['class_name'], "green"))
            for contour in o['vertexes']:
                if use_polygon:
                    coords = ",".join("%.4f,%.4f" % (x + 1, y + 1) for x, y in contour)
                    self.obj_regions.append("polygon(%s) %s" % (coords, meta))
                else:
                    self.obj_regions.append("box(%.4f,%.4f,%.4f,%.4f,0) %s" % (xc + 1, yc + 1, dx, dy, meta))
        return 0

    def write_ds9_regions(self, outfile):
        if not self.obj_regions:
            logger.warning("Region list with detected objects is empty, nothing to be written...")
            return
        with open(outfile, 'w') as fp:
            fp.write("# Region file format: DS9\nglobal color=green\nimage\n")
            for line in self.obj_regions:
                fp.write(line + "\n")
