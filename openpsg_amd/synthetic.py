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
               force_id0=False, tiny_object=False, features=True, device="cpu"):
    """Returns dict(mask_features, pan_results, object_id_list, img_meta, categories)."""
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
