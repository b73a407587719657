"""Class-name tables of the PSG open-vocabulary setting (data, not code).

Values follow the reference's parameter source configs/psg/baseline_v4_ov.py:15-47 (COCO-panoptic
thing/stuff names and the 56 PSG predicates).  The relation head indexes ``object_categories`` with
``object_id % 1000`` (relation_transformer_head_v4.py:138) after the ``-stuff/-merged/-other`` suffixes
have been stripped (mask2former_relation_v2.py:23-35); ``strip_suffix`` below restates that rule.
"""

INSTANCE_OFFSET = 1000  # mmdet.core.INSTANCE_OFFSET: id = category + 1000 * instance (openseed_relation_v2.py:125)

THING_CLASSES = (
    'person', 'bicycle', 'car', 'motorcycle', 'airplane', 'bus', 'train', 'truck', 'boat',
    'traffic light', 'fire hydrant', 'stop sign', 'parking meter', 'bench', 'bird', 'cat', 'dog',
    'horse', 'sheep', 'cow', 'elephant', 'bear', 'zebra', 'giraffe', 'backpack', 'umbrella',
    'handbag', 'tie', 'suitcase', 'frisbee', 'skis', 'snowboard', 'sports ball', 'kite',
    'baseball bat', 'baseball glove', 'skateboard', 'surfboard', 'tennis racket', 'bottle',
    'wine glass', 'cup', 'fork', 'knife', 'spoon', 'bowl', 'banana', 'apple', 'sandwich', 'orange',
    'broccoli', 'carrot', 'hot dog', 'pizza', 'donut', 'cake', 'chair', 'couch', 'potted plant',
    'bed', 'dining table', 'toilet', 'tv', 'laptop', 'mouse', 'remote', 'keyboard', 'cell phone',
    'microwave', 'oven', 'toaster', 'sink', 'refrigerator', 'book', 'clock', 'vase', 'scissors',
    'teddy bear', 'hair drier', 'toothbrush',
)

STUFF_CLASSES = (
    'banner', 'blanket', 'bridge', 'cardboard', 'counter', 'curtain', 'door', 'floor-wood',
    'flower', 'fruit', 'gravel', 'house', 'light', 'mirror', 'net', 'pillow', 'platform',
    'playingfield', 'railroad', 'river', 'road', 'roof', 'sand', 'sea', 'shelf', 'snow', 'stairs',
    'tent', 'towel', 'wall-brick', 'wall-stone', 'wall-tile', 'wall-wood', 'water', 'window-blind',
    'window', 'tree', 'fence', 'ceiling', 'sky', 'cabinet', 'table', 'floor', 'pavement',
    'mountain', 'grass', 'dirt', 'paper', 'food', 'building', 'rock', 'wall', 'rug',
)

RELATION_CLASSES = (
    'over', 'in front of', 'beside', 'on', 'in', 'attached to', 'hanging from', 'on back of',
    'falling off', 'going down', 'painted on', 'walking on', 'running on', 'crossing',
    'standing on', 'lying on', 'sitting on', 'flying over', 'jumping over', 'jumping from',
    'wearing', 'holding', 'carrying', 'looking at', 'guiding', 'kissing', 'eating', 'drinking',
    'feeding', 'biting', 'catching', 'picking', 'playing with', 'chasing', 'climbing', 'cleaning',
    'playing', 'touching', 'pushing', 'pulling', 'opening', 'cooking', 'talking to', 'throwing',
    'slicing', 'driving', 'riding', 'parked on', 'driving on', 'about to hit', 'kicking',
    'swinging', 'entering', 'exiting', 'enclosing', 'leaning on',
)


def strip_suffix(name: str) -> str:
    for suffix in ("-stuff", "-merged", "-other"):
        name = name.replace(suffix, "")
    return name


object_categories = [strip_suffix(n) for n in THING_CLASSES + STUFF_CLASSES]
relation_categories = list(RELATION_CLASSES)

assert len(object_categories) == 133 and len(relation_categories) == 56
