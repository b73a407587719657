"""mmdet registries when mmdet is importable, a minimal work-alike otherwise.

The reference registers its classes by import side effect (configs/psg/baseline_v4_ov.py:7-13
`custom_imports`; `@HEADS.register_module()` at relation_transformer_head_v4.py:20,
`@DETECTORS.register_module()` at openseed_relation_v2.py:19) and builds them from
`dict(type='RelationTransformerHeadV4', ...)` (CFG:58-63, openseed_relation_v2.py:69).
mmdet / mmcv are not installed here, so the shim reproduces exactly that much.
"""
from __future__ import annotations


class Registry:
    def __init__(self, name):
        self.name, self._modules = name, {}

    def register_module(self, name=None, force=False, module=None):
        def _reg(cls):
            key = name or cls.__name__
            if key in self._modules and not force:
                raise KeyError(f"{key} is already registered in {self.name}")
            self._modules[key] = cls
            return cls
        return _reg(module) if module is not None else _reg

    def get(self, key):
        return self._modules.get(key)

    def build(self, cfg: dict, **default_args):
        cfg = dict(cfg)
        typ = cfg.pop("type")
        cls = self.get(typ) if isinstance(typ, str) else typ
        if cls is None:
            raise KeyError(f"{typ} is not in the {self.name} registry")
        for k, v in default_args.items():
            cfg.setdefault(k, v)
        return cls(**cfg)


try:  # pragma: no cover - mmdet is absent in this image
    from mmdet.models.builder import HEADS, DETECTORS, build_head, build_detector  # type: ignore
    HAVE_MMDET = True
except Exception:  # noqa: BLE001
    HAVE_MMDET = False
    HEADS = Registry("head")
    DETECTORS = Registry("detector")

    def build_head(cfg):
        return HEADS.build(cfg)

    def build_detector(cfg, train_cfg=None, test_cfg=None):
        return DETECTORS.build(cfg, train_cfg=train_cfg, test_cfg=test_cfg)
