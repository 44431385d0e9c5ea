"""Minimal OpenMMLab-style registries so the reference's config dicts build
this package's classes unchanged (the plugin boundary
is `type='...'` strings + ctor kwargs).  When a real mmcv/mmdet is importable
the classes are ALSO registered there, so `plugin_dir='transcar_amd/'` works
as a drop-in for `projects/mmdet3d_plugin/`.

Only the names the REFERENCE's plugin owns go into the mm registries (`export=True`: Detr3DHead,
Detr3DTransformer, Detr3DTransformerDecoder, Detr3DCrossAtten, NMSFreeCoder, HungarianAssigner3D,
BBox3DL1Cost -- projects/mmdet3d_plugin/__init__.py:1-16 registers exactly these on this path).  The
third-party names the configs also use (MultiheadAttention, DetrTransformerDecoderLayer, FocalLossCost,
IoUCost: mmcv / mmdet own them) stay in THIS package's registries only -- the head builds its nested
modules through them -- so another model built in the same process keeps mmcv's own classes.
"""
import importlib
import warnings


class Registry:
    def __init__(self, name, mm_path=None):
        self.name = name
        self.module_dict = {}
        self._mm_path = mm_path        # ('module', 'ATTR') of the mm registry

    def _mm_registry(self):
        if not self._mm_path:
            return None
        try:
            mod = importlib.import_module(self._mm_path[0])
            return getattr(mod, self._mm_path[1])
        except Exception:
            return None

    def register_module(self, name=None, force=False, module=None, export=False):
        """`export=True`: a reference-owned name -- also registered in the real mmcv / mmdet registry when one
        is importable, replacing the reference plugin's class of the same name (that is the drop-in)."""
        def _reg(cls):
            key = name or cls.__name__
            if key in self.module_dict and not force:
                raise KeyError('%s is already registered in %s' % (key, self.name))
            self.module_dict[key] = cls
            if export:
                mm = self._mm_registry()
                if mm is not None:
                    try:
                        mm.register_module(name=key, force=True, module=cls)
                    except Exception as e:      # an mm version with another signature: say so, keep going
                        warnings.warn('transcar_amd: could not register %s in %s.%s: %r'
                                      % (key, self._mm_path[0], self._mm_path[1], e))
            return cls
        if module is not None:
            return _reg(module)
        return _reg

    def get(self, key):
        if key not in self.module_dict:
            raise KeyError('%s is not in the %s registry' % (key, self.name))
        return self.module_dict[key]

    def build(self, cfg, **default_args):
        if cfg is None:
            return None
        cfg = dict(cfg)
        for k, v in default_args.items():
            cfg.setdefault(k, v)
        if 'type' not in cfg:
            raise KeyError('cfg for %s needs a "type": %r' % (self.name, cfg))
        typ = cfg.pop('type')
        cls = self.get(typ) if isinstance(typ, str) else typ
        return cls(**cfg)


ATTENTION = Registry('attention', ('mmcv.cnn.bricks.registry', 'ATTENTION'))
TRANSFORMER_LAYER = Registry('transformerLayer',
                             ('mmcv.cnn.bricks.registry', 'TRANSFORMER_LAYER'))
TRANSFORMER_LAYER_SEQUENCE = Registry(
    'transformer-layers sequence',
    ('mmcv.cnn.bricks.registry', 'TRANSFORMER_LAYER_SEQUENCE'))
TRANSFORMER = Registry('Transformer', ('mmdet.models.utils.builder', 'TRANSFORMER'))
HEADS = Registry('heads', ('mmdet.models', 'HEADS'))
BBOX_CODERS = Registry('bbox_coder', ('mmdet.core.bbox.builder', 'BBOX_CODERS'))
BBOX_ASSIGNERS = Registry('bbox_assigner',
                          ('mmdet.core.bbox.builder', 'BBOX_ASSIGNERS'))
MATCH_COST = Registry('Match Cost',
                      ('mmdet.core.bbox.match_costs.builder', 'MATCH_COST'))


def build_attention(cfg, **kw):
    return ATTENTION.build(cfg, **kw)


def build_transformer_layer(cfg, **kw):
    return TRANSFORMER_LAYER.build(cfg, **kw)


def build_transformer_layer_sequence(cfg, **kw):
    return TRANSFORMER_LAYER_SEQUENCE.build(cfg, **kw)


def build_transformer(cfg, **kw):
    return TRANSFORMER.build(cfg, **kw)


def build_head(cfg, **kw):
    return HEADS.build(cfg, **kw)


def build_bbox_coder(cfg, **kw):
    return BBOX_CODERS.build(cfg, **kw)
