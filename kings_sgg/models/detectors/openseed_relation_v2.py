"""Same dotted path as the reference module (configs/psg/baseline_v4_ov.py:10)."""
from openpsg_amd.detector import OpenSeeDRelationV2  # noqa: F401
