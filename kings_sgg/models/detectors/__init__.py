from .openseed_relation_v2 import OpenSeeDRelationV2  # noqa: F401
