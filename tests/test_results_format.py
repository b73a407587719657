"""Submission writer vs the reference's format (tools/infer.py:149-187) - host logic, no GPU."""
import json
import os

import numpy as np

from openpsg_amd.results import render_result, rgb2id, write_submission


def _res():
    pan = np.full((6, 8), 133, dtype=np.int32)
    pan[:3, :4] = 0            # person #0 (aliases void in real outputs)
    pan[3:, 4:] = 1017         # category 17, instance 1
    return dict(pan_results=pan, rel_results=dict(object_id_list=[0, 1017, 133], relation=[[0, 1, 3], [1, 0, 55]]),
                rel_scores=[1, 1])


def test_rgb2id_matches_panopticapi():
    assert rgb2id((1, 2, 3)) == 1 + 256 * 2 + 65536 * 3


def test_submission_files(tmp_path):
    path = write_submission([_res(), dict(pan_results=np.zeros((2, 2), np.int32),
                                          rel_results=dict(object_id_list=[], relation=[]), rel_scores=[])],
                            str(tmp_path))
    data = json.load(open(path))
    assert [d["pan_seg_file_name"] for d in data] == ["0.png", "1.png"]
    first = data[0]
    assert first["relations"] == [[0, 1, 4], [1, 0, 56]]                      # predicate + 1 (INFER:180)
    assert [s["category_id"] for s in first["segments_info"]] == [1, 18]       # id % 1000 + 1; 133 skipped
    assert data[1]["relations"] == [[0, 0, 1]] and len(data[1]["segments_info"]) == 1   # empty padding (INFER:171-176)
    from PIL import Image
    img = np.asarray(Image.open(os.path.join(tmp_path, "submission", "panseg", "0.png")))
    img = img.astype(np.int64)
    ids = img[..., 0] + 256 * img[..., 1] + 65536 * img[..., 2]
    assert ids[0, 0] == first["segments_info"][0]["id"] and ids[5, 7] == first["segments_info"][1]["id"]
    assert ids[0, 7] == 0                                                       # background stays black
