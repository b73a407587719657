"""Run by tests/test_gpu_plans.py in a FRESH process: one image through the fp32s head at Llama-2-7B width (2 layers, K = 20
selected pairs = 960 prompt-pass rows: the shapes llm._SPLIT_PLAN_TABLE names), over generic fp32 weights and over fp16-valued
ones (two-plane prompt pass, 1920 rows), plus three images through the mixed head's forward_batch (60 decode rows:
llm._BATCH_PLAN_TABLE).  Prints one digest per leg of everything a plan could change: existence logits, first-step LLM
logits, greedy tokens."""
import hashlib
import json
import sys

import torch


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(a.detach().cpu().contiguous().numpy().tobytes() if torch.is_tensor(a) else a.tobytes())
    return h.hexdigest()


def main():
    from openpsg_amd.config import LlamaConfig, PSGConfig, QFormerConfig
    from openpsg_amd.head import RelationTransformerHeadV4
    from openpsg_amd.synthetic import make_scene
    from openpsg_amd.weights import make_weights_device
    dev = torch.device("cuda:0")
    cfg = PSGConfig(qformer=QFormerConfig(), llm=LlamaConfig(layers=2), max_object_num=12)
    scene = make_scene((512, 512), 12, seed=5, device="cuda:0")
    inp = dict(mask_features=scene["mask_features"], img_metas=[scene["img_meta"]],
               object_info=[dict(object_id_list=scene["object_id_list"], pan_results=scene["pan_results"])])
    out = {}
    for leg, dtype, vals in (("fp32s", "fp32s", None), ("fp32s_w16", "fp32s", torch.float16), ("mixed_batch", "mixed", None)):
        tdt = torch.float32 if dtype == "fp32s" else torch.float16
        w = make_weights_device(cfg, 3, dev, llm_dtype=tdt, llm_values=vals)
        head = RelationTransformerHeadV4(dtype=dtype, device="cuda:0", tokenizers="word", max_object_num=12, llm_config=cfg.llm,
                                         on_parse_error="skip", suppress_eos=True)
        head.load_weights(w)
        del w
        if leg == "mixed_batch":
            head.forward_batch([inp, inp, inp])
            out[leg] = digest(*[head.last_batch[i]["tokens_host"] for i in range(3)])
        else:
            head(inp)
            torch.cuda.synchronize()
            rows = int(head.last["llm_inputs"].shape[0] * head.last["llm_inputs"].shape[1])
            out[leg] = digest(head.last["exist_logit"], head.last["first_logits"], head.last["tokens"])
            out[leg + "_prompt_rows"] = rows
            if leg == "fp32s_w16":
                out["w16_streamed"] = bool(head.llm_engine._w16_all)
        del head
        torch.cuda.empty_cache()
    print("PLAN_PROBE " + json.dumps(out))


if __name__ == "__main__":
    sys.exit(main())
