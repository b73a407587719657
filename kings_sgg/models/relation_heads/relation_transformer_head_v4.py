"""Same dotted path as the reference module (configs/psg/baseline_v4_ov.py:11); importing it
registers `RelationTransformerHeadV4` in the HEADS registry, as the reference's decorator does."""
from openpsg_amd.categories import object_categories, relation_categories  # noqa: F401
from openpsg_amd.head import RelationTransformerHeadV4  # noqa: F401
