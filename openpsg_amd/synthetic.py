"""Synthetic scenes for parity tests and the benchmark.

A scene is what `OpenSeeDRelationV2.forward_openseed` hands the relation head
(openseed_relation_v2.py:112-143, 177-181): `mask_features [1,256,H/4,W/4]` fp32, a panoptic id
map at ORIGINAL resolution with ids `category + 1000 * instance`, the object id list, and the
three mmdet shapes.  Rectangles are painted in order (later overwrite earlier).
"""
from __future__ import annotations

import numpy as np
import torch

from .categories import INSTANCE_OFFSET


def make_scene(pad_hw, num_objects, seed=0, ori_hw=None, img_hw=None, void_id=133, channels=256,
               force_id0=False, tiny_object=False, features=True, device="cpu", num_categories=133):
    """Returns dict(mask_features, pan_results, object_id_list, img_meta, categories).
    num_categories < 133: the objects draw from that many classes only (real images repeat classes: several
    instances of `person`, ...), so many pairs share a prompt."""
    rng = np.random.default_rng(seed)
    pad_h, pad_w = pad_hw
    img_h, img_w = img_hw if img_hw is not None else pad_hw
    H0, W0 = ori_hw if ori_hw is not None else (img_h, img_w)
    pan = np.full((H0, W0), void_id, dtype=np.int32)
    lo, hi = (96, 384) if max(pad_hw) >= 1024 else (48, 192)
    lo, hi = int(lo * H0 / img_h), int(hi * H0 / img_h)
    counts, ids, cats = {}, [], []
    for k in range(num_objects):
        cat = int(rng.integers(0, 133))
        if num_categories < 133:
            cat = (cat * 7) % 133 % num_categories * (133 // num_categories)   # spread over things and stuff
        if force_id0 and k == 0:
            cat = 0
        inst = counts.get(cat, -1) + 1
        counts[cat] = inst
        oid = cat + INSTANCE_OFFSET * inst
        h, w = int(rng.integers(lo, hi + 1)), int(rng.integers(lo, hi + 1))
        if tiny_object and k == num_objects - 1:
            h = w = 5                                     # misses every stride-64 sample point
        y, x = int(rng.integers(0, max(1, H0 - h))), int(rng.integers(0, max(1, W0 - w)))
        if tiny_object and k == num_objects - 1:
            y, x = y // 64 * 64 + 20, x // 64 * 64 + 20
        pan[y:y + h, x:x + w] = oid
        ids.append(oid)
        cats.append(cat)
    scene = dict(
        pan_results=torch.from_numpy(pan).to(device),
        object_id_list=[torch.tensor(i, dtype=torch.int32) for i in ids],
        categories=cats,
        img_meta=dict(img_shape=(img_h, img_w, 3), pad_shape=(pad_h, pad_w, 3), ori_shape=(H0, W0, 3)),
    )
    if features:
        if device == "cpu":
            f = rng.standard_normal((1, channels, pad_h // 4, pad_w // 4), dtype=np.float32)
            scene["mask_features"] = torch.from_numpy(f)
        else:
            g = torch.Generator(device=device)
            g.manual_seed(seed)
            scene["mask_features"] = torch.randn((1, channels, pad_h // 4, pad_w // 4), generator=g,
                                                 device=device, dtype=torch.float32)
    return scene


class BitmapMasksLike:
    """Stand-in for mmdet's BitmapMasks as the training branch uses it (V4:371: `.to_tensor(dtype, device)`)."""

    def __init__(self, masks: np.ndarray):
        self.masks = masks

    def to_tensor(self, dtype, device):
        return torch.from_numpy(self.masks).to(device=device, dtype=dtype)


def make_train_scene(pad_hw, categories_, gt_rels, seed=0, channels=256, device="cpu"):
    """Training-mode inputs of the head (V4:114-133, 360-406): padded ground-truth thing masks
    [n_thing, pad_h, pad_w] (BitmapMasks), a padded semantic map [1, pad_h, pad_w] holding stuff labels (255 =
    none), `masks_info` [{category, is_thing}] (things: category < 80, COCO panoptic order) and `gt_rels`
    [(subject, object, predicate)].  Rectangles are painted in order."""
    rng = np.random.default_rng(seed)
    H, W = pad_hw
    sem = np.full((H, W), 255, dtype=np.int64)
    thing_masks, info = [], []
    lo, hi = (96, 384) if max(pad_hw) >= 1024 else (48, 192)
    for c in categories_:
        y, x = int(rng.integers(0, H - lo)), int(rng.integers(0, W - lo))
        h, w = int(rng.integers(lo, hi + 1)), int(rng.integers(lo, hi + 1))
        is_thing = c < 80
        if is_thing:
            m = np.zeros((H, W), dtype=np.uint8)
            m[y:y + h, x:x + w] = 1
            thing_masks.append(m)
        else:
            sem[y:y + h, x:x + w] = c
        info.append(dict(category=int(c), is_thing=bool(is_thing)))
    feat = rng.standard_normal((1, channels, H // 4, W // 4), dtype=np.float32)
    masks = np.stack(thing_masks) if thing_masks else np.zeros((0, H, W), dtype=np.uint8)
    meta = dict(masks_info=info, gt_rels=[[tuple(int(v) for v in r) for r in gt_rels]],
                img_shape=(H, W, 3), pad_shape=(H, W, 3), ori_shape=(H, W, 3))
    return dict(mask_features=torch.from_numpy(feat).to(device), img_metas=[meta],
                gt_masks=[BitmapMasksLike(masks)],
                gt_labels=[torch.tensor([c for c in categories_ if c < 80], dtype=torch.long)],
                gt_semantic_seg=[torch.from_numpy(sem)[None].to(device)])
