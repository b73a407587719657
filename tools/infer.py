"""mmdet-free work-alike of the reference's tools/infer.py (INFER:65-188) for the MI355X path.

    python tools/infer.py --segmenter synthetic --images 8 --objects 50 --out work_dirs/demo
    python tools/infer.py --segmenter precomputed --seg-dir seg_npz/ --list files.txt --out work_dirs/run

The segmenter (OpenSeeD in the reference, DET2:92-143) is pluggable: `synthetic` rectangles
 or `.npz` files of precomputed OpenSeeD outputs.  Everything downstream - relation
head on the GPU, result packing (DET2:183-190), submission files (INFER:149-187) - is this repo's.
Random-init weights are used unless --checkpoint / --llm are given (no model files exist offline).
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--segmenter", choices=["synthetic", "precomputed"], default="synthetic")
    ap.add_argument("--seg-dir")
    ap.add_argument("--list", help="text file with one image file name per line (precomputed mode)")
    ap.add_argument("--images", type=int, default=4)
    ap.add_argument("--objects", type=int, default=20)
    ap.add_argument("--size", type=int, nargs=2, default=[1024, 1024], help="pad_shape H W")
    ap.add_argument("--ori-size", type=int, nargs=2, default=None)
    ap.add_argument("--out", default="work_dirs/demo")
    ap.add_argument("--llm-layers", type=int, default=2)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--selector", choices=["topk", "threshold"], default="topk")
    ap.add_argument("--checkpoint", help="reference-style partial checkpoint (state_dict with relation_head.* keys)")
    a = ap.parse_args()

    from openpsg_amd.config import LlamaConfig, PSGConfig, QFormerConfig
    from openpsg_amd.detector import OpenSeeDRelationV2, PrecomputedSegmenter, SyntheticSegmenter
    from openpsg_amd.head import RelationTransformerHeadV4
    from openpsg_amd.results import write_submission
    from openpsg_amd.weights import make_weights_device
    dev = torch.device("cuda:0")
    llm = LlamaConfig(layers=a.llm_layers)
    cfg = PSGConfig(qformer=QFormerConfig(), llm=llm, max_object_num=a.objects)
    head = RelationTransformerHeadV4(dtype=a.dtype, device="cuda:0", tokenizers="word", max_object_num=a.objects,
                                     llm_config=llm, on_parse_error="skip", pair_selector=a.selector)
    head.load_weights(make_weights_device(cfg, 0, dev))
    if a.checkpoint:
        sd = torch.load(a.checkpoint, map_location="cpu")
        sd = sd.get("state_dict", sd)
        head.load_state_dict({k[len("relation_head."):]: v for k, v in sd.items() if k.startswith("relation_head.")},
                             strict=False)
    seg = SyntheticSegmenter(a.objects, seed=0) if a.segmenter == "synthetic" else PrecomputedSegmenter(a.seg_dir)
    det = OpenSeeDRelationV2(relation_head=head, segmenter=seg)
    if a.segmenter == "synthetic":
        names = [f"{i}.jpg" for i in range(a.images)]
    else:
        names = [l.strip() for l in open(a.list) if l.strip()]
    pad = tuple(a.size)
    ori = tuple(a.ori_size) if a.ori_size else pad
    results, t0 = [], time.time()
    for name in names:
        meta = dict(filename=name, ori_shape=ori + (3,), img_shape=pad + (3,), pad_shape=pad + (3,))
        results.append(det.simple_test(None, [meta])[0])
    torch.cuda.synchronize()
    dt = time.time() - t0
    path = write_submission(results, a.out)
    n_rel = sum(len(r["rel_results"]["relation"]) for r in results)
    print(f"{len(names)} images in {dt:.2f}s ({len(names) / dt:.2f} img/s), {n_rel} relations -> {path}")


if __name__ == "__main__":
    main()
