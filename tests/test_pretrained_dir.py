"""The LLM of a local HuggingFace checkpoint directory (V4:99-103: `AutoModelForCausalLM.from_pretrained(llm_model_name)`,
then the first `llm_truncate_num` layers).  CPU: the reader - config, tensor names under `language_model.`, stored dtype,
single file / sharded safetensors / .bin, truncation, refusals.  GPU (`-m gpu`): a head built from the directory decodes
what a head given the same tensors as a dict decodes."""
import json
import os

import numpy as np
import pytest
import torch

from openpsg_amd.config import PSGConfig, QFormerConfig, tiny_llm
from openpsg_amd.weights import (is_hf_checkpoint_dir, llm_shapes, make_weights_numpy, read_hf_llama_config,
                                 read_hf_llama_weights)


def _hf_config(m, **over):
    c = dict(architectures=["LlamaForCausalLM"], hidden_size=m.hidden, num_attention_heads=m.heads,
             num_key_value_heads=m.heads, num_hidden_layers=m.layers, intermediate_size=m.inter, vocab_size=m.vocab,
             rms_norm_eps=m.rms_eps, rope_theta=m.rope_theta, bos_token_id=m.bos, eos_token_id=m.eos,
             torch_dtype="float16", tie_word_embeddings=False)
    c.update(over)
    return c


def _write_dir(path, cfg, w, fmt, dtype=torch.float16):
    """A checkpoint directory as `save_pretrained` leaves it: names WITHOUT the head's `language_model.` prefix."""
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(_hf_config(cfg.llm), f)
    sd = {k[len("language_model."):]: v.to(dtype).contiguous() for k, v in w.items() if k.startswith("language_model.")}
    if fmt == "safetensors":
        from safetensors.torch import save_file
        save_file(sd, os.path.join(path, "model.safetensors"))
    elif fmt == "safetensors_sharded":
        from safetensors.torch import save_file
        names = sorted(sd)
        parts = [names[0::2], names[1::2]]
        wm = {}
        for i, part in enumerate(parts):
            fn = f"model-{i + 1:05d}-of-00002.safetensors"
            save_file({k: sd[k] for k in part}, os.path.join(path, fn))
            wm.update({k: fn for k in part})
        with open(os.path.join(path, "model.safetensors.index.json"), "w") as f:
            json.dump({"metadata": {}, "weight_map": wm}, f)
    else:
        sd["model.layers.0.self_attn.rotary_emb.inv_freq"] = torch.ones(64)      # old checkpoints carry this buffer
        torch.save(sd, os.path.join(path, "pytorch_model.bin"))
    return sd


@pytest.fixture(scope="module")
def tiny():
    cfg = PSGConfig(qformer=QFormerConfig(vocab=512), llm=tiny_llm(256, 3, 512, 512), max_object_num=30)
    return cfg, make_weights_numpy(cfg, seed=5)


@pytest.mark.parametrize("fmt", ["safetensors", "safetensors_sharded", "bin"])
def test_reader_returns_the_reference_names_in_the_stored_dtype(tmp_path, tiny, fmt):
    cfg, w = tiny
    d = str(tmp_path / fmt)
    sd = _write_dir(d, cfg, w, fmt)
    assert is_hf_checkpoint_dir(d) and not is_hf_checkpoint_dir("meta-llama/Llama-2-7b-hf")
    assert read_hf_llama_config(d) == cfg.llm
    got = read_hf_llama_weights(d)
    assert set(got) == set(llm_shapes(cfg))                                   # the head's own names, no stray buffer
    for k, v in got.items():
        assert v.dtype == torch.float16 and tuple(v.shape) == llm_shapes(cfg)[k]
        assert torch.equal(v, sd[k[len("language_model."):]])
    two = read_hf_llama_weights(d, n_layers=2)                                # llm_truncate_num = 2 (V4:101-103)
    assert set(two) == {k for k in got if ".layers.2." not in k}


def test_reader_refuses_what_the_kernels_are_not_built_for(tmp_path, tiny):
    from openpsg_amd._lib import PsgHipError
    cfg, w = tiny
    d = str(tmp_path / "gqa")
    _write_dir(d, cfg, w, "safetensors")
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(_hf_config(cfg.llm, num_key_value_heads=1), f)
    with pytest.raises(PsgHipError, match="grouped-query"):
        read_hf_llama_config(d)
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(_hf_config(cfg.llm, num_attention_heads=4, num_key_value_heads=4), f)
    with pytest.raises(PsgHipError, match="head_dim"):
        read_hf_llama_config(d)
    e = str(tmp_path / "empty")
    os.makedirs(e)
    with open(os.path.join(e, "config.json"), "w") as f:
        json.dump(_hf_config(cfg.llm), f)
    with pytest.raises(PsgHipError, match="no model.safetensors"):
        read_hf_llama_weights(e)
    from openpsg_amd.weights import hf_checkpoint_has_weights
    assert not hf_checkpoint_has_weights(e) and hf_checkpoint_has_weights(d)    # a config / tokenizer-only directory: the head
    # then takes the architecture from it and waits for load_llm_weights()


@pytest.mark.gpu
def test_head_built_from_a_checkpoint_directory_decodes_like_the_head_given_the_tensors(tmp_path):
    """V4:99-103 through the constructor: `llm_model_name` = a local directory holding an fp16 checkpoint.  The head reads
    config and weights itself, keeps 2 of its 3 layers (llm_truncate_num), recognises the fp16 values (streams 2 bytes
    per weight) and gives the tokens and first-step logits of a head that was handed the upcast tensors as a dict; a
    checkpoint of the head loaded AFTERWARDS (language_projection included) reaches the engine's packed projection."""
    from openpsg_amd.head import RelationTransformerHeadV4
    from openpsg_amd.synthetic import make_scene
    cfg = PSGConfig(qformer=QFormerConfig(vocab=512), llm=tiny_llm(256, 3, 512, 512), max_object_num=30)
    w = make_weights_numpy(cfg, seed=5)
    w16 = {k: (v.half().float() if k.startswith("language_model.") else v) for k, v in w.items()}   # what from_pretrained yields
    d = str(tmp_path / "llama")
    _write_dir(d, cfg, w, "safetensors_sharded")
    kw = dict(dtype="fp32s", device="cuda:0", qformer_vocab_size=512, tokenizers="word", max_object_num=30,
              on_parse_error="skip", suppress_eos=True, llm_truncate_num=2, llm_feature_size=256)
    a = RelationTransformerHeadV4(llm_model_name=d, **kw)                         # reads the directory
    assert a.cfg.llm == cfg.llm and a.llm_engine.n_layers == 2 and a.llm_engine._w16_all
    a.load_state_dict({k: v for k, v in w.items() if not k.startswith("language_model.")}, strict=False)
    b = RelationTransformerHeadV4(llm_config=cfg.llm, **kw)
    b.load_weights(w16)
    scene = make_scene((512, 512), 6, seed=3, device="cuda:0")
    inputs = dict(mask_features=scene["mask_features"], img_metas=[scene["img_meta"]],
                  object_info=[dict(object_id_list=scene["object_id_list"], pan_results=scene["pan_results"])])
    ra, rb = a(inputs), b(inputs)
    assert torch.equal(a.last["exist_logit"], b.last["exist_logit"])
    assert torch.equal(a.last["tokens"], b.last["tokens"]) and torch.equal(a.last["first_logits"], b.last["first_logits"])
    assert ra["rel_pred"] == rb["rel_pred"]
    # (a's engine packed an all-zero language_projection at construction: equality with b shows it was refreshed)


@pytest.mark.gpu
@pytest.mark.parametrize("i2", [1, 0])
def test_a_reloaded_language_projection_reaches_every_packed_copy_of_the_engine(i2):
    """The decode engine packs `language_projection` when it is built; in the fp32s mode's row-invariant path (the dealt
    decodes of a pair-sharded job) it also keeps a split image of it - the interleaved one in a cache keyed by the
    tensor's ADDRESS, which the allocator reuses for the replacement.  A checkpoint loaded after the engine exists must
    reach all of them: the head then decodes what a fresh head built on the new weights decodes, bit for bit."""
    from openpsg_amd import _lib
    from openpsg_amd.head import RelationTransformerHeadV4
    from openpsg_amd.synthetic import make_scene
    cfg = PSGConfig(qformer=QFormerConfig(vocab=512), llm=tiny_llm(256, 2, 512, 512), max_object_num=30)
    w1, w2 = make_weights_numpy(cfg, seed=5), make_weights_numpy(cfg, seed=5)
    g = torch.Generator().manual_seed(1)
    w2["language_projection.weight"] = w1["language_projection.weight"] + 0.05 * torch.randn(
        w1["language_projection.weight"].shape, generator=g)
    kw = dict(dtype="fp32s", device="cuda:0", qformer_vocab_size=512, tokenizers="word", max_object_num=30,
              on_parse_error="skip", suppress_eos=True, llm_config=cfg.llm, llm_feature_size=256)
    old = _lib.get_option(0, "split_i2")
    _lib.set_option(0, "split_i2", i2)
    try:
        a = RelationTransformerHeadV4(**kw).load_weights(w1)
        a.llm_engine.row_invariant = True
        scene = make_scene((512, 512), 6, seed=3, device="cuda:0")
        inputs = dict(mask_features=scene["mask_features"], img_metas=[scene["img_meta"]],
                      object_info=[dict(object_id_list=scene["object_id_list"], pan_results=scene["pan_results"])])
        a(inputs)
        first = a.last["first_logits"].clone()
        a.load_state_dict({k: v for k, v in w2.items() if not k.startswith("language_model.")}, strict=False)
        a(inputs)
        b = RelationTransformerHeadV4(**kw).load_weights(w2)
        b.llm_engine.row_invariant = True
        b(inputs)
        assert not torch.equal(first, a.last["first_logits"])                  # the new projection is in use ...
        assert torch.equal(a.last["first_logits"], b.last["first_logits"])     # ... in every copy
        assert torch.equal(a.last["tokens"], b.last["tokens"])
    finally:
        _lib.set_option(0, "split_i2", old)


@pytest.mark.gpu
def test_infer_tool_takes_the_llm_from_a_checkpoint_directory(tmp_path):
    """tools/infer.py --llm-dir: the work-alike of the reference's tools/infer.py builds its head on a local checkpoint
    directory (3 layers on disk, --llm-layers 2 keeps two: llm_truncate_num) and writes a submission."""
    from tools import infer
    cfg = PSGConfig(qformer=QFormerConfig(vocab=512), llm=tiny_llm(256, 3, 512, 512), max_object_num=30)
    d = str(tmp_path / "llama")
    _write_dir(d, cfg, make_weights_numpy(cfg, seed=5), "safetensors")
    a = infer.parser().parse_args(["--images", "2", "--objects", "6", "--size", "512", "512", "--dtype", "fp32s",
                                   "--llm-dir", d, "--llm-layers", "2", "--out", str(tmp_path / "out")])
    head = infer.build_head(a, torch.device("cuda:0"))
    assert head.cfg.llm == cfg.llm and head.llm_engine.n_layers == 2 and head.llm_engine._w16_all
    results, path = infer.run(a, head=head)
    assert len(results) == 2 and os.path.isfile(path)
