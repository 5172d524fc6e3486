"""Datasets as the training/inference drivers see them: the bookkeeping base class of
mrcnn/utils.py:305-453 (class / image registries, ``prepare``) and the radio-source dataset of
scripts/run.py:246-815 (FITS image + one FITS mask per object, listed either as
``image.fits,mask.fits,label`` rows or as caesar JSON files with an ``objs`` list; class dictionary
given on the command line).  Host-side only."""
import json
import logging
import os
import uuid

import numpy as np

from . import fits

logger = logging.getLogger("mrcnn")


class Dataset(object):
    def __init__(self, class_map=None):
        self._image_ids = []
        self.image_info = []
        self.class_info = [{"source": "", "id": 0, "name": "BG"}]
        self.source_class_ids = {}

    def add_class(self, source, class_id, class_name):
        assert "." not in source, "Source name cannot contain a dot"
        for info in self.class_info:
            if info['source'] == source and info["id"] == class_id:
                return
        self.class_info.append({"source": source, "id": class_id, "name": class_name})

    def add_image(self, source, image_id, path, **kwargs):
        info = {"id": image_id, "source": source, "path": path}
        info.update(kwargs)
        self.image_info.append(info)

    def prepare(self, class_map=None):
        self.num_classes = len(self.class_info)
        self.class_ids = np.arange(self.num_classes)
        self.class_names = [",".join(c["name"].split(",")[:1]) for c in self.class_info]
        self.num_images = len(self.image_info)
        self._image_ids = np.arange(self.num_images)
        self.class_from_source_map = {"{}.{}".format(i['source'], i['id']): k
                                      for i, k in zip(self.class_info, self.class_ids)}
        self.sources = list(set(i['source'] for i in self.class_info))
        self.source_class_ids = {}
        for source in self.sources:
            self.source_class_ids[source] = [i for i, info in enumerate(self.class_info)
                                             if i == 0 or source == info['source']]

    @property
    def image_ids(self):
        return self._image_ids

    def image_reference(self, image_id):
        return self.image_info[image_id]["path"]

    def load_mask(self, image_id):
        return np.empty([0, 0, 0]), np.empty([0], np.int32)


class SourceDataset(Dataset):
    """scripts/run.py:246-815."""

    def __init__(self):
        super(SourceDataset, self).__init__()
        self.class_id_map = {}
        self.nclasses = 0
        self.nobjs_per_class = []
        self.convert_to_rgb = True
        self.convert_to_uint8 = True
        self.apply_zscale = True
        self.zscale_contrasts = [0.25, 0.25, 0.25]
        self.apply_biascontrast = False
        self.bias = 0.5
        self.contrast = 1.0

    def set_class_dict(self, class_dict_str):
        """'{"sidelobe":1,"source":2,"galaxy":3}' -> registered classes (run.py:272-318)."""
        try:
            class_dict = json.loads(class_dict_str) if isinstance(class_dict_str, str) else dict(class_dict_str)
        except Exception:
            logger.error("Failed to convert class dict string to dict!")
            return -1
        self.class_id_map = class_dict
        self.class_id_map.setdefault("bkg", 0)
        self.class_info = [{"source": "", "id": 0, "name": "BG"}]
        for name, cid in sorted(class_dict.items(), key=lambda kv: kv[1]):
            if name in ("bkg", "background"):
                continue
            self.add_class("rg-dataset", cid, name)
        self.nclasses = len(self.class_info)
        self.nobjs_per_class = [0] * self.nclasses
        return 0

    def _register(self, img_path, mask_paths, class_ids, **extra):
        self.add_image("rg-dataset", image_id=str(uuid.uuid1()), path=img_path, path_masks=mask_paths,
                       class_ids=class_ids, **extra)
        for c in class_ids:
            if 0 <= c < len(self.nobjs_per_class):
                self.nobjs_per_class[c] += 1

    def load_data_from_list(self, dataset, nmaximgs=-1):
        """Rows ``image.fits,mask.fits,label`` (one object per image; run.py:374-440)."""
        n = 0
        with open(dataset, "r") as f:
            for line in f:
                parts = [p.strip() for p in line.strip().split(",")]
                if len(parts) < 3 or not parts[0]:
                    continue
                img, mask, label = parts[:3]
                if label not in self.class_id_map:
                    logger.warning("Image file %s class name (%s) is not present in dictionary, skip it..." % (img, label))
                    continue
                if not (os.path.isfile(img) and os.path.isfile(mask)):
                    logger.warning("Image or mask of row '%s' does not exist, skip it..." % line.strip())
                    continue
                self._register(os.path.abspath(img), [os.path.abspath(mask)], [self.class_id_map[label]])
                n += 1
                if 0 < nmaximgs <= n:
                    break
        return 0 if n > 0 else -1

    def load_data_from_json_file(self, filename, rootdir='', modify_class_names=True):
        """One caesar JSON file: {"img":..., "objs":[{"mask":..., "class":..., "sidelobe-mixed":...,
        "nislands":...}], metadata...} (run.py:445-551)."""
        try:
            with open(filename, "r") as fh:
                d = json.load(fh)
        except IOError:
            logger.error("Failed to open file %s, skip it..." % filename)
            return -1
        img_full = os.path.abspath(os.path.join(rootdir, d['img']))
        if not (os.path.isfile(img_full) and img_full.endswith('.fits')):
            logger.warning("Image file %s does not exist or has unexpected extension (.fits required)" % img_full)
            return -1
        meta = {k: d.get(k) for k in ("telescope", "bkg", "rms", "bmaj", "bmin", "dx", "dy", "nx", "ny")}
        masks, ids, near = [], [], []
        for obj in d['objs']:
            mask_full = os.path.abspath(os.path.join(rootdir, obj['mask']))
            if not (os.path.isfile(mask_full) and mask_full.endswith('.fits')):
                logger.error("One or more mask of file %s does not exist or have unexpected extension" % img_full)
                return -1
            name = obj['class']
            if modify_class_names:
                if obj.get('nislands', 1) > 1 and name == "extended":
                    name = 'extended-multisland'
                if obj.get('sidelobe-mixed'):
                    name = 'flagged'
                obj['class'] = name
            if name not in self.class_id_map:
                logger.warning("Image file %s class name (%s) is not present in dictionary, skip it..." % (img_full, name))
                continue
            masks.append(mask_full)
            ids.append(self.class_id_map[name])
            near.append(1 if (obj.get('sidelobe-mixed') == 1 or obj.get('sidelobe-near') == 1) else 0)
        self._register(img_full, masks, ids, sidelobes_mixed_or_near=near, objs=d['objs'], metadata=meta)
        return 0

    def load_data_from_json_list(self, filelist, nmaximgs=-1):
        n = 0
        with open(filelist, "r") as f:
            for line in f:
                fn = line.strip()
                if not fn:
                    continue
                if self.load_data_from_json_file(fn, os.path.dirname(fn)) == 0:
                    n += 1
                if 0 < nmaximgs <= n:
                    break
        return 0 if n > 0 else -1

    def load_data_from_json_search(self, topdir, nmaximgs=-1):
        n = 0
        for root, _, files in sorted(os.walk(topdir)):
            for fn in sorted(files):
                if fn.endswith(".json") and self.load_data_from_json_file(os.path.join(root, fn), root) == 0:
                    n += 1
                    if 0 < nmaximgs <= n:
                        return 0
        return 0 if n > 0 else -1

    def load_mask(self, image_id):
        info = self.image_info[image_id]
        if info["source"] != "rg-dataset":
            return super(SourceDataset, self).load_mask(image_id)
        mask = None
        for k, fn in enumerate(info["path_masks"]):
            data, _ = fits.read_fits(fn, stretch=False, normalize=False, convertToRGB=False)
            if mask is None:
                mask = np.zeros([data.shape[0], data.shape[1], len(info["path_masks"])], dtype=bool)
            mask[:, :, k] = data.astype(bool)
        return mask, np.asarray(info["class_ids"], dtype=np.int32)

    def load_gt_masks(self, image_id, binary=True):
        m, _ = self.load_mask(image_id)
        return m if binary else m.astype(int)

    def load_image(self, image_id):
        # self.device (set by MaskRCNN.train when Config.DEVICE_FITS): NaN fill / zscale / RGB on that GPU, same bytes back
        image, _ = fits.read_fits(self.image_info[image_id]['path'], stretch=self.apply_zscale,
                                  zscale_contrasts=self.zscale_contrasts, normalize=True,
                                  convertToRGB=self.convert_to_rgb, to_uint8=self.convert_to_uint8,
                                  stretch_biascontrast=self.apply_biascontrast, bias=self.bias, contrast=self.contrast,
                                  device=getattr(self, "device", None))
        return image

    def image_uuid(self, image_id):
        return self.image_info[image_id]["id"]

    def compute_class_weights(self):
        tot = float(sum(self.nobjs_per_class[1:])) or 1.0
        return {i: (tot / (len(self.nobjs_per_class) - 1) / n if n > 0 else 1.0)
                for i, n in enumerate(self.nobjs_per_class)}
