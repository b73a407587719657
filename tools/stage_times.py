"""Stage timings of the hot path on one MI355X (development aid; bench.py is the contract)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpsg_amd.config import LlamaConfig, PSGConfig, QFormerConfig  # noqa: E402
from openpsg_amd.head import RelationTransformerHeadV4  # noqa: E402
from openpsg_amd.synthetic import make_scene  # noqa: E402
from openpsg_amd.weights import make_weights_device  # noqa: E402
from openpsg_amd.categories import INSTANCE_OFFSET, object_categories  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--objects", type=int, default=50)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--llm-layers", type=int, default=32)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    llm = LlamaConfig(layers=a.llm_layers)
    cfg = PSGConfig(qformer=QFormerConfig(), llm=llm, max_object_num=a.objects)
    t0 = time.time()
    w = make_weights_device(cfg, 0, dev, llm_dtype=torch.bfloat16 if a.dtype == "bf16" else torch.float32)
    torch.cuda.synchronize()
    print(f"weights in {time.time() - t0:.1f}s, mem {torch.cuda.memory_allocated() / 2**30:.1f} GiB", flush=True)
    head = RelationTransformerHeadV4(dtype=a.dtype, device="cuda:0", tokenizers="word", max_object_num=a.objects,
                                     on_parse_error="skip", suppress_eos=True)
    head.load_weights(w)
    del w
    torch.cuda.empty_cache()
    scene = make_scene((a.size, a.size), a.objects, seed=0, device="cuda:0")
    obj_ids = [int(i) for i in scene["object_id_list"]]
    names = [object_categories[i % INSTANCE_OFFSET] for i in obj_ids]
    feat, meta, pan = scene["mask_features"], scene["img_meta"], scene["pan_results"]
    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
    for it in range(a.iters):
        e = [ev() for _ in range(4)]
        e[0].record()
        rq = head.run_relation_query(feat, meta, obj_ids, names, pan)
        e[1].record()
        out = head.decode_selected(rq, names)
        e[2].record()
        torch.cuda.synchronize()
        print(f"iter {it}: relation-query {e[0].elapsed_time(e[1]):.2f} ms, decode {e[1].elapsed_time(e[2]):.2f} ms, "
              f"pairs/s full {a.objects * (a.objects - 1) / (e[0].elapsed_time(e[2]) / 1e3):.0f}", flush=True)
    print("peak mem GiB", torch.cuda.max_memory_allocated() / 2**30)


if __name__ == "__main__":
    main()
