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


def load_case(name):
    g = dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
    llm = tiny_llm(int(g["llm_hidden"]), int(g["llm_layers"]), int(g["llm_inter"]), int(g["llm_vocab"]))
    cfg = PSGConfig(qformer=QFormerConfig(vocab=512), llm=llm, max_object_num=30)
    w = make_weights_numpy(cfg, seed=int(g["weight_seed"]))
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
