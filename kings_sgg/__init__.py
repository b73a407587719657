"""Import-path shim: the reference's configs register classes by importing
`kings_sgg.models.relation_heads.relation_transformer_head_v4` and
`kings_sgg.models.detectors.openseed_relation_v2` (configs/psg/baseline_v4_ov.py:7-13).
These modules re-export the MI355X-native implementations from `openpsg_amd`."""
