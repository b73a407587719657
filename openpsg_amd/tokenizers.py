"""Word-level stand-ins for the two HF tokenizers the reference head calls.

The reference builds its prompts from strings and tokenises them with the InstructBLIP Q-Former
(BERT) tokenizer, right padded (relation_transformer_head_v4.py:146-152), and with the Llama-2
tokenizer, LEFT padded with pad = unk (V4:104-105, 260-266), then `batch_decode`s generated ids
(V4:313).  Neither vocabulary file exists offline, so tests / goldens / the benchmark use this
deterministic word-level tokenizer with the same call surface (`__call__(..., padding=True)`,
`padding_side`, `batch_decode`, `pad_token`/`unk_token`).  At deployment the head accepts any HF
tokenizer object instead.
"""
from __future__ import annotations

import re

import torch

from .categories import object_categories, relation_categories

_WORD = re.compile(r"[A-Za-z0-9\-]+|[^\sA-Za-z0-9]")


def default_words():
    words = set()
    for name in list(object_categories) + list(relation_categories):
        words.update(_WORD.findall(name.lower()))
    for s in ("Is there a relation between {} and {}?",
              "What are the relations between {} and {}? Assistant: "):
        words.update(w for w in _WORD.findall(s.lower()) if w not in "{}")
    return sorted(words)


class WordTokenizer:
    """style='bert': [CLS] w.. [SEP], right pad (id 0).  style='llama': <s> w.., left/right pad (id 0 = <unk>)."""

    def __init__(self, style: str, words=None):
        assert style in ("bert", "llama")
        self.style = style
        words = default_words() if words is None else list(words)
        if style == "bert":
            self.specials = ["[PAD]", "[UNK]", "[CLS]", "[SEP]"]
            self.padding_side = "right"
        else:
            self.specials = ["<unk>", "<s>", "</s>"]
            self.padding_side = "right"       # HF default; the reference sets 'left' before use (V4:262)
        self.id_to_piece = self.specials + words
        self.piece_to_id = {p: i for i, p in enumerate(self.id_to_piece)}
        self.unk_token = "[UNK]" if style == "bert" else "<unk>"
        self.pad_token = "[PAD]" if style == "bert" else None
        self.unk_id = self.piece_to_id[self.unk_token]

    @property
    def vocab_size(self):
        return len(self.id_to_piece)

    @property
    def pad_id(self):
        return self.piece_to_id[self.pad_token if self.pad_token is not None else self.unk_token]

    def encode(self, text: str):
        ids = [self.piece_to_id.get(w, self.unk_id) for w in _WORD.findall(text.lower())]
        if self.style == "bert":
            return [self.piece_to_id["[CLS]"]] + ids + [self.piece_to_id["[SEP]"]]
        return [self.piece_to_id["<s>"]] + ids

    def __call__(self, texts, return_tensors="pt", padding=True, return_attention_mask=True):
        if isinstance(texts, str):
            texts = [texts]
        enc = [self.encode(t) for t in texts]
        T = max(len(e) for e in enc)
        ids = torch.full((len(enc), T), self.pad_id, dtype=torch.long)
        mask = torch.zeros((len(enc), T), dtype=torch.long)
        for r, e in enumerate(enc):
            if self.padding_side == "left":
                ids[r, T - len(e):] = torch.tensor(e)
                mask[r, T - len(e):] = 1
            else:
                ids[r, :len(e)] = torch.tensor(e)
                mask[r, :len(e)] = 1
        return {"input_ids": ids, "attention_mask": mask}

    def decode(self, ids):
        return " ".join(self.id_to_piece[int(i)] if int(i) < len(self.id_to_piece) else self.unk_token
                        for i in ids)

    def batch_decode(self, sequences):
        return [self.decode(s) for s in sequences]
