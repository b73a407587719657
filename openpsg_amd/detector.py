"""`OpenSeeDRelationV2` shell: the caller of the relation head.

Mirrors kings_sgg/models/detectors/openseed_relation_v2.py (DET2): registry name (DET2:19-20),
constructor keywords (DET2:21-31), `simple_test(imgs, img_metas)` result packing (DET2:170-190) and
the OpenSeeD -> mmdet id conversion (DET2:112-132).  OpenSeeD itself (un-vendored fork +
detectron2, DET2:13-16) is out of scope; a `Segmenter` supplies what `openseed.forward` returns.
"""
from __future__ import annotations

import os
from typing import Protocol

import numpy as np
import torch
import torch.nn as nn

from .categories import INSTANCE_OFFSET
from .registry import DETECTORS, build_head
from .synthetic import make_scene


class Segmenter(Protocol):
    """What DET2:107 (`self.openseed.forward(batch_inputs)`) provides for one image."""

    def __call__(self, img: torch.Tensor, img_meta: dict):
        """-> (panoptic_seg [H0,W0] int tensor, segments_info [{'id','category_id'}...],
               mask_features [1,256,pad_h/4,pad_w/4] fp32)"""
        ...


def panoptic_to_mmdet(panoptic_seg: torch.Tensor, segments_info):
    """DET2:112-132.  id = category + 1000 * (k-th instance of that category); the map is
    zero-initialised (DET2:114), so unassigned pixels alias id 0 == the first 'person'."""
    out = torch.zeros_like(panoptic_seg, dtype=torch.int32)
    seen, object_id_list = {}, []
    for seg in segments_info:
        cat = int(seg["category_id"])
        seen[cat] = seen.get(cat, -1) + 1
        oid = cat + INSTANCE_OFFSET * seen[cat]
        out[panoptic_seg == int(seg["id"])] = oid
        object_id_list.append(torch.tensor(oid, dtype=torch.int32))
    return out, object_id_list


class SyntheticSegmenter:
    """Seeded rectangles in OpenSeeD's output format."""

    def __init__(self, num_objects=10, seed=0, device="cuda", seed_from_filename=False):
        """seed_from_filename: image `<k>.jpg` always gets scene seed + k, whichever rank processes it and in whatever
        order (images dealt to several ranks, tools/infer.py); otherwise the seed advances by one per call."""
        self.num_objects, self.seed, self.device = num_objects, seed, device
        self.base_seed, self.seed_from_filename = seed, seed_from_filename

    def __call__(self, img, img_meta):
        pad = img_meta["pad_shape"][:2]
        seed = self.seed
        stem = os.path.splitext(os.path.basename(str(img_meta.get("filename", ""))))[0]
        if self.seed_from_filename and stem.isdigit():
            seed = self.base_seed + int(stem)
        s = make_scene(pad, self.num_objects, seed=seed, ori_hw=img_meta["ori_shape"][:2],
                       img_hw=img_meta["img_shape"][:2], device=self.device)
        self.seed += 1
        ids = [int(i) for i in s["object_id_list"]]
        # re-express as OpenSeeD segments: segment id = 1..N, background 0
        pan = s["pan_results"]
        seg = torch.zeros_like(pan)
        info = []
        for k, oid in enumerate(ids):
            seg[pan == oid] = k + 1
            info.append(dict(id=k + 1, category_id=oid % INSTANCE_OFFSET))
        return seg, info, s["mask_features"]


class PrecomputedSegmenter:
    """Reads `<dir>/<stem>.npz` with arrays panoptic_seg, segment_ids, category_ids, mask_features
    (the ingest format for OpenSeeD outputs computed elsewhere)."""

    def __init__(self, directory, device="cuda"):
        self.directory, self.device = directory, device

    def __call__(self, img, img_meta):
        stem = os.path.splitext(os.path.basename(img_meta["filename"]))[0]
        z = np.load(os.path.join(self.directory, stem + ".npz"))
        info = [dict(id=int(i), category_id=int(c)) for i, c in zip(z["segment_ids"], z["category_ids"])]
        return (torch.from_numpy(z["panoptic_seg"]).to(self.device), info,
                torch.from_numpy(z["mask_features"]).to(self.device))


class OpenSeeDSegmenter:
    """Adapter around a LIVE OpenSeeD model (the un-vendored fork the reference builds at DET2:36-41: any object whose
    `forward(batch_inputs)` returns `(outputs, mask_features)` with `outputs[0]['panoptic_seg'] = (id map, segments_info)`
    and whose `.model.pixel_mean / .pixel_std` hold the normalisation constants).  Does what DET2:94-109 does around that
    call: the mmdet-normalised image back to 0..255, the padding removed (`img_shape`), OpenSeeD told the original size
    (`ori_shape`).  A deployment that has OpenSeeD installed passes `segmenter=OpenSeeDSegmenter(openseed)`."""

    def __init__(self, openseed):
        self.openseed = openseed

    @torch.no_grad()
    def __call__(self, img, img_meta):
        model = self.openseed.model
        mean = model.pixel_mean.clone().to(img.device).view(3, 1, 1)                      # DET2:98-100
        std = model.pixel_std.clone().to(img.device).view(3, 1, 1)
        img = img * std + mean
        h, w = img_meta['img_shape'][:2]                                                 # DET2:102-103: no padding
        img = img[:, :h, :w]
        batch_inputs = [{'image': img, 'height': img_meta['ori_shape'][0], 'width': img_meta['ori_shape'][1]}]
        outputs, mask_features = self.openseed.forward(batch_inputs)                     # DET2:107
        seg, info = outputs[0]['panoptic_seg']
        return seg, info, mask_features


@DETECTORS.register_module()
class OpenSeeDRelationV2(nn.Module):
    def __init__(self, openseed_config_path='', openseed_pretrained_path='', thing_classes=(), stuff_classes=(),
                 relation_head=None, train_cfg=None, test_cfg=None, init_cfg=None, segmenter: Segmenter = None,
                 **kwargs):
        super().__init__()
        self.thing_classes, self.stuff_classes = list(thing_classes), list(stuff_classes)
        self.segmenter = segmenter
        self.relation_head = build_head(relation_head) if isinstance(relation_head, dict) else relation_head
        self.freeze_layers = list((train_cfg or {}).get('freeze_layers', []))           # DET2:72-79
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.train(False)

    def forward_openseed(self, imgs, img_metas, mode='test'):
        assert imgs is None or imgs.size(0) == 1, 'only support batch size 1'                # DET2:93
        if self.segmenter is None:
            raise RuntimeError("no segmenter: OpenSeeD is not vendored; pass segmenter=SyntheticSegmenter(...) "
                               "or PrecomputedSegmenter(dir)")
        meta = img_metas[0]
        seg, info, mask_features = self.segmenter(None if imgs is None else imgs[0], meta)
        pan_results, object_id_list = panoptic_to_mmdet(seg, info)
        result = {'pan_results': pan_results, 'object_id_list': object_id_list,
                  'object_score_list': [torch.tensor(1.0) for _ in info], 'ins_results': None}   # DET2:131-132
        return [result], mask_features

    @torch.no_grad()
    def simple_test(self, imgs, img_metas, **kwargs):
        """DET2:170-190."""
        results, mask_features = self.forward_openseed(imgs, img_metas, mode='test')
        head_out = self.relation_head(dict(mask_features=mask_features, img_metas=img_metas, object_info=results))
        return [self._pack(results[0], head_out)]

    @staticmethod
    def _pack(res, head_out):
        """DET2:183-188."""
        res['pan_results'] = res['pan_results'].detach().cpu().numpy()
        res['rel_results'] = dict(object_id_list=[oid.item() for oid in res['object_id_list']],
                                  relation=head_out['rel_pred'])
        res['rel_scores'] = head_out['rel_score']
        return res

    @torch.no_grad()
    def simple_test_submit(self, imgs, img_metas, slot=0):
        """`simple_test` issued one image ahead (RelationTransformerHeadV4.submit): segmenter + head enqueued on the
        slot's HIP stream; returns a callable whose call waits for that stream and gives simple_test's result."""
        results, mask_features = self.forward_openseed(imgs, img_metas, mode='test')
        pending = self.relation_head.submit(dict(mask_features=mask_features, img_metas=img_metas, object_info=results),
                                            slot=slot)
        return lambda: [self._pack(results[0], pending.result())]

    @torch.no_grad()
    def simple_test_batch(self, imgs_list, img_metas_list, **kwargs):
        """Throughput mode: simple_test for several images whose selected pairs are decoded together
        (RelationTransformerHeadV4.forward_batch).  One entry per image, each as simple_test takes it."""
        segs = [self.forward_openseed(imgs, metas, mode='test') for imgs, metas in zip(imgs_list, img_metas_list)]
        outs = self.relation_head.forward_batch(
            [dict(mask_features=feat, img_metas=metas, object_info=results)
             for (results, feat), metas in zip(segs, img_metas_list)])
        return [[self._pack(results[0], out)] for (results, _), out in zip(segs, outs)]

    def forward_train(self, img, img_metas, gt_bboxes=None, gt_labels=None, gt_masks=None, gt_semantic_seg=None,
                      gt_bboxes_ignore=None, **kwargs):
        """DET2:145-168: the (frozen) segmenter's mask features + the ground truth go to the relation head, whose two
        losses come back with their gradient graph (fp32 head; openpsg_amd/train_graph.py).  The segmenter's own losses
        are empty: it is frozen (DET2:72-79) and not part of this build."""
        with torch.no_grad():
            _, mask_features = self.forward_openseed(img, img_metas, mode='train')
        losses = {}
        losses.update(self.relation_head(dict(mask_features=mask_features, img_metas=img_metas, gt_labels=gt_labels,
                                              gt_masks=gt_masks, gt_semantic_seg=gt_semantic_seg)))
        return losses

    def forward(self, img=None, img_metas=None, return_loss=False, **kwargs):
        if return_loss:
            return self.forward_train(img, img_metas, **kwargs)
        return self.simple_test(img, img_metas, **kwargs)
