"""Image -> (`img` tensor, `img_metas`) for the mmdet-free driver.

The reference delegates this to mmdet 2.x's test pipeline and only configures it
(configs/psg/baseline_v4_ov.py:109-123: LoadImageFromFile, MultiScaleFlipAug(flip=False) around
Resize(keep_ratio=True), Normalize, Pad(size_divisor=32), Collect(['img'])); tools/infer.py:39-41
overrides `img_scale` to (1333, 1333).  What the relation head consumes of it is the three shapes
`ori_shape`, `img_shape`, `pad_shape` (relation_transformer_head_v4.py:416-419;
openseed_relation_v2.py:102-105 uses them to strip the padding for the segmenter).

mmdet / mmcv are absent from this image, so their arithmetic is restated from the published
mmcv 1.x behaviour (`imrescale`: factor = min(long_cap / long_edge, short_cap / short_edge), new
size = int(edge * factor + 0.5); bilinear; `impad_to_multiple` pads right/bottom with zeros).  The
shape arithmetic is exact integer work and is tested; the resampling itself (cv2 INTER_LINEAR in
mmcv, torch bilinear with half-pixel centres here) only feeds the segmenter, which is out of scope.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

IMG_NORM_MEAN = (123.675, 116.28, 103.53)      # CFG:74-75, RGB order (to_rgb=True)
IMG_NORM_STD = (58.395, 57.12, 57.375)


def rescale_size(h: int, w: int, scale=(1333, 1333)):
    """mmcv.imrescale's size rule for a (long, short) cap."""
    long_cap, short_cap = max(scale), min(scale)
    factor = min(long_cap / max(h, w), short_cap / min(h, w))
    return int(h * float(factor) + 0.5), int(w * float(factor) + 0.5), factor


def pad_size(h: int, w: int, divisor: int = 32):
    return -(-h // divisor) * divisor, -(-w // divisor) * divisor


def image_meta(ori_hw, scale=(1333, 1333), divisor: int = 32, filename: str = ""):
    """The `img_metas` entry the pipeline would produce for an image of `ori_hw`, without touching pixels."""
    h0, w0 = int(ori_hw[0]), int(ori_hw[1])
    h1, w1, _ = rescale_size(h0, w0, scale)
    hp, wp = pad_size(h1, w1, divisor)
    sf = np.array([w1 / w0, h1 / h0, w1 / w0, h1 / h0], dtype=np.float32)
    return dict(filename=filename, ori_filename=filename, ori_shape=(h0, w0, 3), img_shape=(h1, w1, 3),
                pad_shape=(hp, wp, 3), scale_factor=sf, flip=False, flip_direction=None,
                img_norm_cfg=dict(mean=np.array(IMG_NORM_MEAN, dtype=np.float32),
                                  std=np.array(IMG_NORM_STD, dtype=np.float32), to_rgb=True))


def preprocess_image(img_rgb_u8, scale=(1333, 1333), divisor: int = 32, filename: str = "", device="cpu"):
    """img_rgb_u8: [H, W, 3] uint8 RGB (numpy or tensor).  -> (img [1,3,pad_h,pad_w] fp32 normalised, [img_meta])."""
    x = torch.as_tensor(np.asarray(img_rgb_u8)).to(device)
    assert x.dim() == 3 and x.shape[2] == 3, "expected an HxWx3 image"
    meta = image_meta(x.shape[:2], scale, divisor, filename)
    h1, w1 = meta["img_shape"][:2]
    hp, wp = meta["pad_shape"][:2]
    x = x.permute(2, 0, 1)[None].float()
    if (h1, w1) != tuple(x.shape[-2:]):
        x = F.interpolate(x, size=(h1, w1), mode="bilinear", align_corners=False)
    mean = torch.tensor(IMG_NORM_MEAN, device=x.device).view(1, 3, 1, 1)
    std = torch.tensor(IMG_NORM_STD, device=x.device).view(1, 3, 1, 1)
    x = (x - mean) / std
    x = F.pad(x, (0, wp - w1, 0, hp - h1), value=0.0)
    return x, [meta]


def load_image(path: str):
    """RGB uint8 array.  (mmcv loads BGR and converts with to_rgb=True; the result is the same RGB image.)"""
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"))
