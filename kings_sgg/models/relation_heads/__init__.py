from .relation_transformer_head_v4 import RelationTransformerHeadV4  # noqa: F401
