"""mmdet-free work-alike of the reference's tools/infer.py (INFER:65-188) for the MI355X path.

    python tools/infer.py --segmenter synthetic --images 8 --objects 50 --out work_dirs/demo
    python tools/infer.py --segmenter precomputed --seg-dir seg_npz/ --list files.txt --image-dir imgs/ --out work_dirs/run

The segmenter (OpenSeeD in the reference, DET2:92-143) is pluggable: `synthetic` rectangles
 or `.npz` files of precomputed OpenSeeD outputs.  Everything downstream - image ->
`img_metas` (CFG:109-123 with INFER:39-41's 1333 scale; openpsg_amd/preprocess.py), relation head on
the GPU, result packing (DET2:183-190), submission files (INFER:149-187; `--keep-scores` = the
tools/predict.py:91-97 variant) - is this repo's.  Random-init weights are used unless
--checkpoint / --llm are given (no model files exist offline).

`run(args, head=None)` is the importable entry (tests inject a head with known weights).
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--segmenter", choices=["synthetic", "precomputed"], default="synthetic")
    ap.add_argument("--seg-dir")
    ap.add_argument("--list", help="text file with one image file name per line (precomputed mode)")
    ap.add_argument("--image-dir", help="directory of the image files: shapes come from the files through the "
                                        "test pipeline (resize to 1333 keep-ratio, pad to 32)")
    ap.add_argument("--images", type=int, default=4)
    ap.add_argument("--objects", type=int, default=20)
    ap.add_argument("--size", type=int, nargs=2, default=[1024, 1024], help="pad_shape H W (no --image-dir)")
    ap.add_argument("--ori-size", type=int, nargs=2, default=None,
                    help="original H W; with it and without --size the pipeline's shape arithmetic is applied")
    ap.add_argument("--out", default="work_dirs/demo")
    ap.add_argument("--llm-layers", type=int, default=2)
    ap.add_argument("--dtype", default="mixed", help="mixed | fp16 | bf16 | fp32 | fp32s (RelationTransformerHeadV4 dtype; fp32s = the reference's precision)")
    ap.add_argument("--selector", choices=["topk", "threshold"], default="topk")
    ap.add_argument("--batch", type=int, default=1,
                    help="images per head call; > 1 decodes their selected pairs together (throughput mode)")
    ap.add_argument("--in-flight", type=int, default=1,
                    help="with --batch 1: images in flight per GPU (2 = image k+1 submitted before image k's result is "
                         "taken, head.submit: 1.19x images per second, same results)")
    ap.add_argument("--keep-scores", action="store_true", help="tools/predict.py:91-97 output variant")
    ap.add_argument("--checkpoint", help="reference-style partial checkpoint (state_dict with relation_head.* keys)")
    ap.add_argument("--llm-dir", help="local HuggingFace checkpoint directory of the LLM (config.json + model.safetensors / "
                    "shards / pytorch_model*.bin): read by the head's constructor as the reference's from_pretrained does "
                    "(V4:99-103); without it the LLM is seeded random weights of --llm-layers layers")
    return ap


def build_head(a, dev):
    from openpsg_amd.config import LlamaConfig, PSGConfig, QFormerConfig
    from openpsg_amd.head import RelationTransformerHeadV4
    from openpsg_amd.weights import make_weights_device
    if getattr(a, "llm_dir", None):
        # the LLM of a real checkpoint directory; --llm-layers > 0 keeps its first layers (llm_truncate_num, V4:101-103)
        from openpsg_amd.weights import read_hf_llama_config
        llm = read_hf_llama_config(a.llm_dir)
        trunc = a.llm_layers if 0 < a.llm_layers < llm.layers else -1
        head = RelationTransformerHeadV4(dtype=a.dtype, device=str(dev), tokenizers="word", max_object_num=a.objects,
                                         llm_model_name=a.llm_dir, llm_feature_size=llm.hidden, llm_truncate_num=trunc,
                                         on_parse_error="skip", pair_selector=a.selector)
        cfg = PSGConfig(qformer=QFormerConfig(), llm=llm, max_object_num=a.objects)
        head.load_weights(make_weights_device(cfg, 0, dev, with_llm=False))
    else:
        llm = LlamaConfig(layers=a.llm_layers)
        cfg = PSGConfig(qformer=QFormerConfig(), llm=llm, max_object_num=a.objects)
        head = RelationTransformerHeadV4(dtype=a.dtype, device=str(dev), tokenizers="word", max_object_num=a.objects,
                                         llm_config=llm, on_parse_error="skip", pair_selector=a.selector)
        head.load_weights(make_weights_device(cfg, 0, dev))
    if a.checkpoint:
        sd = torch.load(a.checkpoint, map_location="cpu")
        sd = sd.get("state_dict", sd)
        head.load_state_dict({k[len("relation_head."):]: v for k, v in sd.items() if k.startswith("relation_head.")},
                             strict=False)
    head.warm_prompts()                 # all class-pair prompts tokenised once: no image waits for a tokenizer
    return head


def image_metas(a, names):
    """img_metas per image: from the files (--image-dir), from --ori-size through the pipeline's shape
    arithmetic, or the fixed --size grid."""
    from openpsg_amd.preprocess import image_meta, load_image
    metas = []
    for n in names:
        if a.image_dir:
            img = load_image(os.path.join(a.image_dir, n))
            metas.append(image_meta(img.shape[:2], filename=n))
        elif a.ori_size and not a.size_given:
            metas.append(image_meta(tuple(a.ori_size), filename=n))
        else:
            pad = tuple(a.size)
            ori = tuple(a.ori_size) if a.ori_size else pad
            metas.append(dict(filename=n, ori_shape=ori + (3,), img_shape=pad + (3,), pad_shape=pad + (3,)))
    return metas


def run_local(a, head, world, rank, dev):
    """This rank's share of the images (dealt round-robin, dist.shard_images) through the detector.
    Returns (image names, [(image index, result)], seconds)."""
    from openpsg_amd.detector import OpenSeeDRelationV2, PrecomputedSegmenter, SyntheticSegmenter
    from openpsg_amd.dist import shard_images
    if not hasattr(a, "size_given"):
        a.size_given = True
    seg = SyntheticSegmenter(a.objects, seed=0, device=str(dev), seed_from_filename=True) \
        if a.segmenter == "synthetic" else PrecomputedSegmenter(a.seg_dir, device=str(dev))
    det = OpenSeeDRelationV2(relation_head=head, segmenter=seg)
    if a.segmenter == "synthetic" and not a.list:
        names = [f"{i}.jpg" for i in range(a.images)]
    else:
        names = [l.strip() for l in open(a.list) if l.strip()]
    all_metas = image_metas(a, names)
    mine = shard_images(len(names), world, rank)
    local_results, t0 = [], time.time()
    if a.batch == 1 and getattr(a, "in_flight", 1) > 1:
        # two images in flight (head.submit): image k+1 is enqueued before image k's result is taken
        pend = []
        for k, i in enumerate(mine):
            pend.append((i, det.simple_test_submit(None, [all_metas[i]], slot=k % a.in_flight)))
            if len(pend) >= a.in_flight:
                j, take = pend.pop(0)
                local_results.append((j, take()[0]))
        for j, take in pend:
            local_results.append((j, take()[0]))
        torch.cuda.synchronize()
        return names, local_results, time.time() - t0
    for b0 in range(0, len(mine), a.batch):
        idx = mine[b0:b0 + a.batch]
        if a.batch == 1:
            outs = [det.simple_test(None, [all_metas[idx[0]]])]
        else:
            outs = det.simple_test_batch([None] * len(idx), [[all_metas[i]] for i in idx])
        local_results += [(i, o[0]) for i, o in zip(idx, outs)]
    torch.cuda.synchronize()
    return names, local_results, time.time() - t0


def finish(a, names, results, world, dt):
    from openpsg_amd.results import write_submission
    path = write_submission(results, a.out, keep_scores=a.keep_scores, names=names if a.keep_scores else None)
    n_rel = sum(len(r["rel_results"]["relation"]) for r in results)
    print(f"{len(names)} images on {world} GPU(s) in {dt:.2f}s ({len(names) / dt:.2f} img/s), {n_rel} relations "
          f"-> {path}")
    return path


def run(a, head=None):
    from openpsg_amd.dist import gather_image_results
    # one process per GPU (torch.distributed.run): whole images are dealt round-robin to the ranks
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    own_group = False
    if world > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo")                   # host objects only; no device collective on this path
            own_group = True
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if head is None:
        head = build_head(a, dev)
    names, local_results, dt = run_local(a, head, world, rank, dev)
    results = gather_image_results(local_results, len(names))
    path = finish(a, names, results, world, dt) if rank == 0 else None
    if own_group:
        import torch.distributed as dist
        dist.destroy_process_group()
    return results, path


def run_fake_world(a, head, world, dev="cuda:0"):
    """The same job over `world` ranks of ONE process, one after the other on one GPU (SURVEY 4 "fake world"): what
    `torch.distributed.run --nproc-per-node world tools/infer.py` computes, without the processes."""
    from openpsg_amd.dist import merge_image_results
    dev = torch.device(dev)
    parts, dt, names = [], 0.0, None
    for rank in range(world):
        names, local_results, t = run_local(a, head, world, rank, dev)
        parts.append(local_results)
        dt = max(dt, t)
    results = merge_image_results(parts, len(names))
    return results, finish(a, names, results, world, dt)


def main():
    ap = parser()
    a = ap.parse_args()
    a.size_given = any(x == "--size" for x in sys.argv)
    run(a)


if __name__ == "__main__":
    main()
