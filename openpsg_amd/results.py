"""On-disk formats either side of the path.

* `write_submission`: the reference's submission format (tools/infer.py:149-187): one panoptic
  PNG per image with a random colour per object (cv2 writes BGR, so the stored pixel is (b,g,r)
  and the segment id is rgb2id((r,g,b)) = r + 256 g + 65536 b, tools/parse_predict.py:16-21),
  `segments_info = [{category_id: id % 1000 + 1, id}]`, `relations = [[s, o, r + 1]]`, empty lists
  padded exactly as INFER:171-176, and one `relation.json` list for the split.
* `save_segmenter_output` / `PrecomputedSegmenter` (detector.py): the ingest format for OpenSeeD
  outputs computed elsewhere (what DET2:107-143 hands the head).
"""
from __future__ import annotations

import json
import os
import random

import numpy as np

from .categories import INSTANCE_OFFSET


def rgb2id(color) -> int:
    return int(color[0]) + 256 * int(color[1]) + 256 * 256 * int(color[2])


def render_result(res: dict, rng: random.Random):
    """One `simple_test` result -> (png array [H,W,3] uint8 in RGB order, segments_info, relations)."""
    pan = np.asarray(res["pan_results"])
    object_id_list = res["rel_results"]["object_id_list"]
    relation = [list(map(int, t)) for t in res["rel_results"]["relation"]]
    png = np.zeros(pan.shape + (3,), dtype=np.int64)
    segments_info = []
    for oid in object_id_list:
        if oid == 133:                                             # background (INFER:154-156)
            continue
        r, g, b = (rng.choice(range(0, 255)) for _ in range(3))
        mask = (pan == oid)[..., None].astype(np.int64)
        png = png + mask * np.array([r, g, b]).reshape(1, 1, 3)    # accumulates on overlap, as the reference does
        segments_info.append(dict(category_id=int(oid % INSTANCE_OFFSET + 1), id=rgb2id((r, g, b))))
    if len(relation) == 0:
        relation = [[0, 0, 0]]
    if len(segments_info) == 0:
        r, g, b = (rng.choice(range(0, 255)) for _ in range(3))
        segments_info = [dict(category_id=1, id=rgb2id((r, g, b)))]
    return png.astype(np.uint8), segments_info, [[s, o, r + 1] for s, o, r in relation]


def write_submission(results, output_dir: str, seed: int = 0, keep_scores: bool = False, names=None,
                     entries=None) -> str:
    """results: list of `OpenSeeDRelationV2.simple_test(...)[0]` dicts, in test order.

    keep_scores / names / entries give the score-preserving variant of tools/predict.py:60-108: every record
    also carries `relation_scores` (:97), starts from the dataset entry `entries[i]` (:92) and the panoptic PNG
    is named after the image (`names[i]`, :84-85) instead of its index (tools/infer.py:168-169)."""
    from PIL import Image
    panseg_dir = os.path.join(output_dir, "submission", "panseg")
    os.makedirs(panseg_dir, exist_ok=True)
    rng = random.Random(seed)
    all_results = []
    for idx, res in enumerate(results):
        png, segments_info, relations = render_result(res, rng)
        stem = str(idx) if names is None else os.path.splitext(os.path.basename(str(names[idx])))[0]
        Image.fromarray(png, mode="RGB").save(os.path.join(panseg_dir, f"{stem}.png"))
        rec = dict(entries[idx]) if entries is not None else {}
        rec.update(relations=relations, segments_info=segments_info)
        if keep_scores:
            rec["relation_scores"] = list(res.get("rel_scores", []))
        else:
            rec["pan_seg_file_name"] = f"{stem}.png"
        all_results.append(rec)
    path = os.path.join(output_dir, "submission", "relation.json")
    with open(path, "w") as f:
        json.dump(all_results, f, default=str)
    return path


def save_segmenter_output(path: str, panoptic_seg, segment_ids, category_ids, mask_features):
    """Ingest format read by `PrecomputedSegmenter`: what OpenSeeD returned for one image."""
    np.savez_compressed(path, panoptic_seg=np.asarray(panoptic_seg), segment_ids=np.asarray(segment_ids),
                        category_ids=np.asarray(category_ids), mask_features=np.asarray(mask_features))
