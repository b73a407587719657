"""One-off robustness sweep: random image sets (0-40 objects, 1-8 images, random geometries) through head.forward_batch and
head.submit in fp32 / fp32s / mixed - no exception, and in the exact fp32 mode every image's triplets equal its own forward().
    python tools/batch_fuzz.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpsg_amd.config import PSGConfig, QFormerConfig, tiny_llm
from openpsg_amd.head import RelationTransformerHeadV4
from openpsg_amd.synthetic import make_scene
from openpsg_amd.weights import make_weights_device

def inputs(s):
    return dict(mask_features=s["mask_features"], img_metas=[s["img_meta"]],
                object_info=[dict(object_id_list=s["object_id_list"], pan_results=s["pan_results"])])
cfg = PSGConfig(qformer=QFormerConfig(vocab=30522), llm=tiny_llm(512, 2, 1024, 512), max_object_num=40)
bad = 0
for dtype in ("fp32", "fp32s", "mixed"):
    head = RelationTransformerHeadV4(dtype=dtype, device="cuda:0", llm_config=cfg.llm, llm_feature_size=512, tokenizers="word",
                                     max_object_num=40, on_parse_error="skip", suppress_eos=(dtype != "mixed"))
    head.load_weights(make_weights_device(cfg, 7, torch.device("cuda:0"), llm_dtype=torch.float32))
    rng = np.random.default_rng(5)
    for it in range(25):
        nimg = int(rng.integers(1, 9))
        scenes = []
        for m in range(nimg):
            n = int(rng.choice([0, 1, 2, 3, 5, 9, 17, 33, 40]))
            pad = (64 * int(rng.integers(4, 19)), 64 * int(rng.integers(4, 19)))
            s = make_scene(pad, max(n, 1), seed=100 * it + m, device="cuda:0", num_categories=int(rng.integers(2, 134)))
            if n == 0:
                s["object_id_list"] = []
            scenes.append(s)
        try:
            single = [head(inputs(s)) for s in scenes]
            toks = []
            outs = head.forward_batch([inputs(s) for s in scenes])
            torch.cuda.synchronize()
            assert len(outs) == nimg
            if dtype == "fp32":
                for i in range(nimg):
                    if outs[i] != single[i]:
                        bad += 1
                        print("MISMATCH", dtype, it, i, [len(s["object_id_list"]) for s in scenes])
            # two in flight over the same scenes
            pend = [head.submit(inputs(s), slot=k % 2) for k, s in enumerate(scenes[:2])]
            res = [p.result() for p in pend]
            for i, r in enumerate(res):
                if dtype == "fp32" and r != single[i]:
                    bad += 1
                    print("SUBMIT MISMATCH", it, i)
        except Exception as e:
            bad += 1
            print("EXC", dtype, it, [len(s["object_id_list"]) for s in scenes], type(e).__name__, str(e)[:300])
    print(dtype, "done")
print("bad =", bad)
