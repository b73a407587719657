"""Shape/parameter contract of the relation-query + LMM-decode path.

Mirrors the constructor arguments of the reference head
(kings_sgg/models/relation_heads/relation_transformer_head_v4.py:22-45) and the two third-party
configs it instantiates (InstructBlipQFormerConfig at :78-82, Llama-2-7b at :99-100).  Defaults are
the reference's; tests shrink vocabularies / LLM width, never the Q-Former geometry the kernels are
specialised for (hidden 768 = 12 heads x 64, 33 query rows).
"""
from dataclasses import dataclass, field, asdict


@dataclass(frozen=True)
class QFormerConfig:
    hidden: int = 768          # qformer_feature_size (V4:28)
    heads: int = 12            # InstructBlipQFormerConfig default
    layers: int = 2            # qformer_layer_num (V4:27)
    inter: int = 3072
    vocab: int = 30522
    max_pos: int = 512
    enc_hidden: int = 256      # object_feature_size (V4:42), cross-attention K/V input width
    ln_eps: float = 1e-12
    num_query: int = 32        # relation_query rows (V4:87-88); +1 rel_cls_query row (V4:89-90)
    hidden_dropout: float = 0.1    # InstructBlipQFormerConfig defaults, which V4:78-84 leaves untouched: active when the
    attn_dropout: float = 0.1      # reference trains (training branch only; inference has no dropout)

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads

    @property
    def q_rows(self) -> int:
        return self.num_query + 1


@dataclass(frozen=True)
class LlamaConfig:
    hidden: int = 4096         # llm_feature_size (V4:37)
    heads: int = 32
    layers: int = 32           # llm_truncate_num keeps the first n (V4:101-103)
    inter: int = 11008
    vocab: int = 32000
    rms_eps: float = 1e-5
    rope_theta: float = 10000.0
    bos: int = 1
    eos: int = 2
    pad: int = 0               # pad_token = unk_token (V4:105) -> id 0 for Llama-2

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads


@dataclass(frozen=True)
class PSGConfig:
    qformer: QFormerConfig = field(default_factory=QFormerConfig)
    llm: LlamaConfig = field(default_factory=LlamaConfig)
    patch_size: int = 16               # V4:26
    feat_channels: int = 256           # object_feature_size (V4:42)
    max_object_num: int = 30           # V4:44 (BASELINE configs need 50/100 -> real parameter)
    max_new_tokens: int = 16           # V4:308
    num_selected: int = 20             # V4:237
    # 'uniform': an empty pair mask gives a uniform softmax over all L patches (additive finfo.min,
    # legacy/eager semantics).  'unmasked': additive -10000 variant.
    empty_row_policy: str = "uniform"

    def to_dict(self):
        return asdict(self)


def tiny_llm(hidden=256, layers=2, inter=512, vocab=512) -> LlamaConfig:
    """A small full-arithmetic Llama for parity tests (head_dim stays 128)."""
    assert hidden % 128 == 0
    return LlamaConfig(hidden=hidden, heads=hidden // 128, layers=layers, inter=inter, vocab=vocab)
