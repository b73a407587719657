"""State-dict schema of the relation head and deterministic weight generators.

Key names are the reference head's own ``state_dict()`` names (prefix ``relation_head.`` inside a
detector checkpoint): timm ``PatchEmbed.proj``, HF ``InstructBlipQFormerModel``,
``relation_query`` / ``rel_cls_query`` (V4:87-90), ``binary_rel_cls_pred`` (V4:92),
``language_projection`` (V4:97-98) and HF ``LlamaForCausalLM`` under ``language_model.``.

Reference checkpoints are partial (part_checkpoint_hook.py:96-116 drops ``language_model.*``), so
loaders must be ``strict=False``; ``llm_keys`` / ``head_keys`` split the schema accordingly.

There are no model files offline, so tests and the benchmark use seeded random weights:
``make_weights_numpy`` (numpy PCG64, sorted-key order, bit-reproducible on any box; used for
parity tests and goldens) and ``make_weights_device`` (torch generator on the GPU; used for the
7B-shaped benchmark where numpy would take minutes).
"""
from __future__ import annotations

import numpy as np
import torch

from .config import PSGConfig


def head_shapes(cfg: PSGConfig) -> dict:
    q = cfg.qformer
    C, P = cfg.feat_channels, cfg.patch_size
    s = {
        "patch_embed.proj.weight": (C, C, P, P),
        "patch_embed.proj.bias": (C,),
        "relation_qformer.embeddings.word_embeddings.weight": (q.vocab, q.hidden),
        "relation_qformer.embeddings.position_embeddings.weight": (q.max_pos, q.hidden),
        "relation_qformer.embeddings.layernorm.weight": (q.hidden,),
        "relation_qformer.embeddings.layernorm.bias": (q.hidden,),
        "relation_query": (1, q.num_query, q.hidden),
        "rel_cls_query": (1, 1, q.hidden),
        "binary_rel_cls_pred.weight": (1, q.hidden),
        "binary_rel_cls_pred.bias": (1,),
        "language_projection.weight": (cfg.llm.hidden, q.hidden),
        "language_projection.bias": (cfg.llm.hidden,),
    }
    for l in range(q.layers):
        p = f"relation_qformer.encoder.layer.{l}."
        for att, kin in (("attention", q.hidden), ("crossattention", q.enc_hidden)):
            s[p + f"{att}.attention.query.weight"] = (q.hidden, q.hidden)
            s[p + f"{att}.attention.query.bias"] = (q.hidden,)
            s[p + f"{att}.attention.key.weight"] = (q.hidden, kin)
            s[p + f"{att}.attention.key.bias"] = (q.hidden,)
            s[p + f"{att}.attention.value.weight"] = (q.hidden, kin)
            s[p + f"{att}.attention.value.bias"] = (q.hidden,)
            s[p + f"{att}.output.dense.weight"] = (q.hidden, q.hidden)
            s[p + f"{att}.output.dense.bias"] = (q.hidden,)
            s[p + f"{att}.output.LayerNorm.weight"] = (q.hidden,)
            s[p + f"{att}.output.LayerNorm.bias"] = (q.hidden,)
        for inter, out in (("intermediate", "output"), ("intermediate_query", "output_query")):
            s[p + f"{inter}.dense.weight"] = (q.inter, q.hidden)
            s[p + f"{inter}.dense.bias"] = (q.inter,)
            s[p + f"{out}.dense.weight"] = (q.hidden, q.inter)
            s[p + f"{out}.dense.bias"] = (q.hidden,)
            s[p + f"{out}.LayerNorm.weight"] = (q.hidden,)
            s[p + f"{out}.LayerNorm.bias"] = (q.hidden,)
    return s


def llm_shapes(cfg: PSGConfig) -> dict:
    m = cfg.llm
    s = {
        "language_model.model.embed_tokens.weight": (m.vocab, m.hidden),
        "language_model.model.norm.weight": (m.hidden,),
        "language_model.lm_head.weight": (m.vocab, m.hidden),
    }
    for l in range(m.layers):
        p = f"language_model.model.layers.{l}."
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            s[p + f"self_attn.{n}.weight"] = (m.hidden, m.hidden)
        s[p + "mlp.gate_proj.weight"] = (m.inter, m.hidden)
        s[p + "mlp.up_proj.weight"] = (m.inter, m.hidden)
        s[p + "mlp.down_proj.weight"] = (m.hidden, m.inter)
        s[p + "input_layernorm.weight"] = (m.hidden,)
        s[p + "post_attention_layernorm.weight"] = (m.hidden,)
    return s


def all_shapes(cfg: PSGConfig) -> dict:
    s = head_shapes(cfg)
    s.update(llm_shapes(cfg))
    return s


def _std_for(key: str, shape) -> tuple[float, float]:
    """(mean, std) per tensor.  HF's default std=0.02 makes every softmax near-uniform, which would
    hide masking / ordering bugs; q/k get a larger std so attention is peaky."""
    if key.endswith("LayerNorm.weight") or key.endswith("layernorm.weight") or key.endswith("norm.weight"):
        return 1.0, 0.1
    if key.endswith(".bias"):
        return 0.0, 0.05
    if key in ("relation_query", "rel_cls_query"):
        return 0.0, 1.0                       # torch.randn in the reference (V4:87-90)
    if "embeddings.weight" in key or key.endswith("embed_tokens.weight"):
        return 0.0, 1.0
    fan_in = int(np.prod(shape[1:]))
    if key.endswith("patch_embed.proj.weight"):
        return 0.0, 1.0 / np.sqrt(fan_in)     # features ~N(0,1) -> patches ~N(0,1)
    if ".attention.query." in key or ".attention.key." in key:
        return 0.0, 2.2 / np.sqrt(fan_in)
    if "q_proj" in key or "k_proj" in key:
        return 0.0, 2.0 / np.sqrt(fan_in)
    if key.endswith("lm_head.weight"):
        return 0.0, 3.0 / np.sqrt(fan_in)     # logit std ~3 -> argmax margins >> fp32 noise
    if key.startswith("binary_rel_cls_pred"):
        return 0.0, 2.0 / np.sqrt(fan_in)
    return 0.0, 1.0 / np.sqrt(fan_in)


def make_weights_numpy(cfg: PSGConfig, seed: int = 0, with_llm: bool = True) -> dict:
    """fp32 CPU tensors, numpy PCG64, filled in sorted-key order (bit-reproducible)."""
    shapes = all_shapes(cfg) if with_llm else head_shapes(cfg)
    rng = np.random.default_rng(seed)
    out = {}
    for key in sorted(shapes):
        mean, std = _std_for(key, shapes[key])
        a = rng.standard_normal(shapes[key], dtype=np.float32) * np.float32(std) + np.float32(mean)
        out[key] = torch.from_numpy(a.astype(np.float32))
    return out


def extend_llm_weights_numpy(w: dict, cfg: PSGConfig, seed_base: int = 1000, threads: int = 8) -> dict:
    """`w` plus every `language_model.*` tensor of `cfg` it lacks (the layers behind a truncated model's), fp32 CPU
    tensors.  ONE numpy PCG64 generator PER TENSOR, seeded `seed_base + i` with i = the tensor's index in the sorted
    schema of `cfg`: bit-reproducible on any box whatever the thread count, and fast enough for the un-truncated
    Llama-2-7B shape (27 GB; `make_weights_numpy`'s single stream would take minutes).  The CPU oracle and the GPU head
    of the 32-layer decode parity check (bench.py `parity...decode_7b_32_layers`, tests/test_gpu_llm7b.py) are both
    fed from this dict, so they hold the same values."""
    from concurrent.futures import ThreadPoolExecutor
    out = dict(w)
    todo = [(i, key, shp) for i, (key, shp) in enumerate(sorted(llm_shapes(cfg).items())) if key not in out]

    def fill(job):                                          # numpy releases the GIL while it draws
        i, key, shp = job
        mean, std = _std_for(key, shp)
        arr = np.random.default_rng(seed_base + i).standard_normal(shp, dtype=np.float32)
        arr *= np.float32(std)
        arr += np.float32(mean)
        return key, torch.from_numpy(arr)
    with ThreadPoolExecutor(max_workers=max(1, int(threads))) as ex:
        out.update(ex.map(fill, todo))
    return out


def llm_matrices_as_fp16_values(w: dict, in_place=()) -> dict:
    """A dict in which every `language_model.*` matrix is the fp32 image of its fp16 rounding - the values the
    reference's frozen fp16 Llama-2-7b-hf checkpoint has after `from_pretrained` upcast it (V4:99-100,
    configs/psg/baseline_v4_ov.py:61-65).  Norm vectors and the head's own tensors are passed through.  Keys named in
    `in_place` are rounded inside their storage (the 27 GB of an un-truncated model are not held twice); every other
    matrix is a new tensor, so a dict sharing tensors with `w` keeps its values."""
    out, in_place = {}, set(in_place)
    for k, v in w.items():
        if k.startswith("language_model.") and v.dim() >= 2:
            if k in in_place:
                v.copy_(v.half().float())
                out[k] = v
            else:
                out[k] = v.half().float()
        else:
            out[k] = v
    return out


def make_weights_device(cfg: PSGConfig, seed: int, device, head_dtype=torch.float32,
                        llm_dtype=torch.bfloat16, with_llm: bool = True, llm_values=None) -> dict:
    """Random-init weights generated directly in HBM (benchmark use: 7B-shaped LLM).
    llm_values=torch.float16 with an fp32 `llm_dtype`: the LLM's matrices hold fp16 VALUES in fp32 tensors - what
    `from_pretrained` makes of the fp16 Llama-2-7b-hf checkpoint the reference loads and freezes (V4:99-100,
    configs/psg/baseline_v4_ov.py:61-65)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = {}
    shapes = all_shapes(cfg) if with_llm else head_shapes(cfg)
    for key in sorted(shapes):
        mean, std = _std_for(key, shapes[key])
        dt = llm_dtype if key.startswith("language_model.") else head_dtype
        t = torch.empty(shapes[key], device=device, dtype=torch.float32 if len(shapes[key]) < 2 else dt)
        if t.dtype == torch.float32:
            t.normal_(mean, std, generator=g)
        else:
            # fill in fp32 chunks to keep the distribution exact, then cast
            flat = t.view(-1)
            step = 1 << 26
            for o in range(0, flat.numel(), step):
                n = min(step, flat.numel() - o)
                chunk = torch.empty(n, device=device).normal_(mean, std, generator=g)
                if llm_values is not None and key.startswith("language_model."):
                    chunk = chunk.to(llm_values).float()
                flat[o:o + n] = chunk.to(dt)
        if t.dtype == torch.float32 and len(shapes[key]) >= 2 and llm_values is not None and key.startswith("language_model."):
            t = t.to(llm_values).float()
        out[key] = t.to(dt) if len(shapes[key]) >= 2 else t
    return out


# ---- the LLM of a local HuggingFace checkpoint directory (V4:99-103) --------------------------------------------------
# The reference builds its LLM with `AutoModelForCausalLM.from_pretrained(llm_model_name, low_cpu_mem_usage=True)`
# (V4:99-100): no dtype, so the fp16 tensors on disk are upcast to fp32; `llm_truncate_num` then keeps the first layers
# (V4:101-103).  There is no hub here and no need for the HF module: the head reads the directory itself - config.json,
# then `model.safetensors` / its sharded index / `pytorch_model*.bin` - and hands the tensors, under the names they have
# inside the reference head (`language_model.` + the checkpoint's own names), to the decode engine, which upcasts them on
# the device.  Tensors stay in their STORED dtype on the host: an fp16 checkpoint is then recognised as such by the
# engine's per-tensor round-trip check (llm.py, option llm_w16) exactly as its fp32 upcast would be.
def is_hf_checkpoint_dir(path) -> bool:
    import os
    return isinstance(path, (str, os.PathLike)) and os.path.isfile(os.path.join(path, "config.json"))


def hf_checkpoint_has_weights(path) -> bool:
    """True when the directory holds a model file `read_hf_llama_weights` can read (a tokenizer-only directory does not)."""
    import os
    return any(os.path.isfile(os.path.join(path, f)) for f in
               ("model.safetensors.index.json", "model.safetensors", "pytorch_model.bin.index.json", "pytorch_model.bin"))


def read_hf_llama_config(path):
    """LlamaConfig of the checkpoint directory `path` (its config.json).  Raises PsgHipError for an architecture the
    decode kernels are not built for (grouped-query attention, head_dim != 128, tied embeddings without an lm_head)."""
    import json
    import os
    from .config import LlamaConfig
    from ._lib import PsgHipError
    with open(os.path.join(path, "config.json")) as f:
        c = json.load(f)
    heads = int(c["num_attention_heads"])
    kv = int(c.get("num_key_value_heads") or heads)
    if kv != heads:
        raise PsgHipError(f"{path}: num_key_value_heads={kv} != num_attention_heads={heads} (grouped-query attention is "
                          "not built; the reference's LLM is Llama-2-7b, multi-head)")
    hidden = int(c["hidden_size"])
    if hidden % heads or hidden // heads != 128:
        raise PsgHipError(f"{path}: head_dim {hidden / heads:g} unsupported (kernels are built for 128)")
    if c.get("rope_scaling"):
        raise PsgHipError(f"{path}: rope_scaling={c['rope_scaling']!r} is not built (Llama-2 has none)")

    def tok(name, default):
        v = c.get(name, default)
        return int(v[0] if isinstance(v, (list, tuple)) else default if v is None else v)
    return LlamaConfig(hidden=hidden, heads=heads, layers=int(c["num_hidden_layers"]), inter=int(c["intermediate_size"]),
                       vocab=int(c["vocab_size"]), rms_eps=float(c.get("rms_norm_eps", 1e-5)),
                       rope_theta=float(c.get("rope_theta", 10000.0)), bos=tok("bos_token_id", 1),
                       eos=tok("eos_token_id", 2), pad=0)


def read_hf_llama_weights(path, n_layers=None, prefix="language_model."):
    """{prefix + name: tensor (stored dtype, CPU)} of the LlamaForCausalLM checkpoint in directory `path`; layers at or
    past `n_layers` (llm_truncate_num, V4:101-103) are not read.  Formats: model.safetensors, model.safetensors.index.json
    + shards, pytorch_model.bin, pytorch_model.bin.index.json + shards (in that order, as from_pretrained prefers them)."""
    import json
    import os
    from ._lib import PsgHipError

    def wanted(name):
        if name.endswith("rotary_emb.inv_freq"):                       # a buffer old checkpoints carry; recomputed (HF-LL:115-128)
            return False
        if n_layers is not None and n_layers > 0 and name.startswith("model.layers."):
            return int(name.split(".")[2]) < n_layers
        return True

    def shards(index_name, single_name):
        idx = os.path.join(path, index_name)
        if os.path.isfile(idx):
            with open(idx) as f:
                wm = json.load(f)["weight_map"]
            files = {}
            for name, fn in wm.items():
                if wanted(name):
                    files.setdefault(fn, []).append(name)
            return files
        if os.path.isfile(os.path.join(path, single_name)):
            return {single_name: None}
        return None

    out = {}
    files = shards("model.safetensors.index.json", "model.safetensors")
    if files is not None:
        from safetensors import safe_open
        for fn, names in sorted(files.items()):
            with safe_open(os.path.join(path, fn), framework="pt", device="cpu") as f:
                for name in (names if names is not None else f.keys()):
                    if wanted(name):
                        out[prefix + name] = f.get_tensor(name)
    else:
        files = shards("pytorch_model.bin.index.json", "pytorch_model.bin")
        if files is None:
            raise PsgHipError(f"{path}: no model.safetensors / pytorch_model.bin (or their .index.json) found")
        for fn, names in sorted(files.items()):
            sd = torch.load(os.path.join(path, fn), map_location="cpu", weights_only=True)
            for name in (names if names is not None else list(sd)):
                if wanted(name):
                    out[prefix + name] = sd[name]
            del sd
    if prefix + "lm_head.weight" not in out:
        raise PsgHipError(f"{path}: no lm_head.weight (tied embeddings are not Llama-2's layout)")
    return out
