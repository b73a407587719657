"""Shared test helpers: rebuild the exact inputs a golden case was captured with."""
import ast
import os

import numpy as np
import torch

from openpsg_amd.categories import object_categories, INSTANCE_OFFSET
from openpsg_amd.config import PSGConfig, QFormerConfig, tiny_llm
from openpsg_amd.synthetic import make_scene
from openpsg_amd.tokenizers import WordTokenizer
from openpsg_amd.weights import make_weights_numpy

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
QFORMER_INSTRUCTION = "Is there a relation between {} and {}?"
LLM_INSTRUCTION = "What are the relations between {} and {}? Assistant: "


_WEIGHT_CACHE = {}


def _case_weights(cfg, seed, name):
    """The PCG64 weights of a golden case.  The 7B-width cases take 20+ s to draw and are loaded by eight test modules:
    keep ONE master copy per process and hand out clones (tests round / reload weights in place)."""
    if name not in _WEIGHT_CACHE:
        w = make_weights_numpy(cfg, seed=seed)
        if sum(v.numel() * v.element_size() for v in w.values()) < (64 << 20):
            return w                                               # small cases: drawing them again is cheaper than a cache
        _WEIGHT_CACHE[name] = w
    return {k: v.clone() for k, v in _WEIGHT_CACHE[name].items()}


def load_case(name):
    g = dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
    llm = tiny_llm(int(g["llm_hidden"]), int(g["llm_layers"]), int(g["llm_inter"]), int(g["llm_vocab"]))
    cfg = PSGConfig(qformer=QFormerConfig(vocab=512), llm=llm, max_object_num=30)
    w = _case_weights(cfg, int(g["weight_seed"]), name)
    scene_kw = ast.literal_eval(str(g["scene_kw"]))
    scene = make_scene(**scene_kw)
    assert np.array_equal(scene["pan_results"].numpy(), g["pan_results"])
    return g, cfg, w, scene


def object_names(scene):
    return [object_categories[int(i) % INSTANCE_OFFSET] for i in scene["object_id_list"]]


def qformer_prompts(scene):
    names = object_names(scene)
    n = len(names)
    tok = WordTokenizer("bert")
    enc = tok([QFORMER_INSTRUCTION.format(names[i // n], names[i % n]) for i in range(n * n)])
    return enc["input_ids"], enc["attention_mask"]


def llm_prompts(scene, selected):
    names = object_names(scene)
    n = len(names)
    tok = WordTokenizer("llama")
    tok.padding_side = "left"
    enc = tok([LLM_INSTRUCTION.format(names[i // n], names[i % n]) for i in selected])
    return enc["input_ids"], enc["attention_mask"]


def unpack_bits(bits, n):
    return torch.from_numpy(np.unpackbits(bits, axis=-1, bitorder="little")[..., :n].astype(bool))


# ---- a Llama whose greedy decode is known in advance (product-output tests) -----------------------------------
class ChainTokenizer(WordTokenizer):
    """WordTokenizer('llama') + one extra piece ';' that decodes to the empty string, so that joining pieces
    with single spaces yields the DOUBLE space the reference splits relation names on (V4:317)."""

    def __init__(self):
        from openpsg_amd.tokenizers import default_words
        super().__init__("llama", default_words() + [";"])

    def decode(self, ids):
        return " ".join("" if self.id_to_piece[int(i)] == ";" else self.id_to_piece[int(i)] for i in ids)


def rig_llm_chain(w, cfg, tok, chain=("over", ";", "in", "front", "of", "</s>")):
    """Overwrites the LLM tensors of `w` so that greedy decoding after the prompt's last token ':' emits `chain`:
    attention and MLP outputs are zeroed (o_proj = down_proj = 0), embeddings are one-hot, norms are 1, and the
    lm_head maps each token to its successor (everything outside the chain -> '</s>').  Returns the token ids."""
    m = cfg.llm
    assert tok.vocab_size <= m.hidden and tok.vocab_size <= m.vocab
    ids = [tok.piece_to_id[p] for p in chain]
    emb = torch.zeros(m.vocab, m.hidden)
    for t in range(min(m.vocab, m.hidden)):
        emb[t, t] = 1.0
    head = torch.zeros(m.vocab, m.hidden)
    head[m.eos, :] = 0.5                                      # default successor: '</s>'
    prev = tok.piece_to_id[":"]
    for t in ids:
        head[:, prev] = 0.0
        head[t, prev] = 1.0
        prev = t
    w["language_model.model.embed_tokens.weight"] = emb
    w["language_model.lm_head.weight"] = head
    w["language_model.model.norm.weight"] = torch.ones(m.hidden)
    for l in range(m.layers):
        p = f"language_model.model.layers.{l}."
        w[p + "self_attn.o_proj.weight"] = torch.zeros(m.hidden, m.hidden)
        w[p + "mlp.down_proj.weight"] = torch.zeros(m.hidden, m.inter)
        w[p + "input_layernorm.weight"] = torch.ones(m.hidden)
        w[p + "post_attention_layernorm.weight"] = torch.ones(m.hidden)
    return ids


# ---- training-branch cases (tests/golden/T*.npz) ------------------------------------------------------------------
def load_train_case(name):
    from openpsg_amd.synthetic import make_train_scene
    g = dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
    llm = tiny_llm(int(g["llm_hidden"]), int(g["llm_layers"]), int(g["llm_inter"]), int(g["llm_vocab"]))
    cfg = PSGConfig(qformer=QFormerConfig(vocab=512), llm=llm, max_object_num=30)
    w = _case_weights(cfg, int(g["weight_seed"]), name)
    inputs = make_train_scene(tuple(int(v) for v in g["pad_hw"]), [int(c) for c in g["categories"]],
                              [tuple(int(v) for v in r) for r in g["gt_rels"]], seed=int(g["scene_seed"]))
    return g, cfg, w, inputs


def train_prompts(inputs):
    """BERT prompts of all pairs and the Llama prompt / label tokenisers the training branch uses (V4:146-152,
    260-281): prompts left-padded, labels right-padded."""
    info = inputs["img_metas"][0]["masks_info"]
    names = [object_categories[x["category"]] for x in info]
    n = len(names)
    enc = WordTokenizer("bert")([QFORMER_INSTRUCTION.format(names[i // n], names[i % n]) for i in range(n * n)])

    def llm_prompt(selected):
        tok = WordTokenizer("llama")
        tok.padding_side = "left"
        e = tok([LLM_INSTRUCTION.format(names[i // n], names[i % n]) for i in selected])
        return e["input_ids"], e["attention_mask"]

    def llm_label(labels):
        tok = WordTokenizer("llama")
        tok.padding_side = "right"
        e = tok(labels)
        return e["input_ids"], e["attention_mask"]
    return enc["input_ids"], enc["attention_mask"], llm_prompt, llm_label
