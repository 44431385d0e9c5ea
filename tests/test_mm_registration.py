"""The OpenMMLab registration path (projects/mmdet3d_plugin/__init__.py:1-16 is the plugin boundary): with
(fake) mmcv / mmdet registries importable, importing this package must put exactly the REFERENCE-OWNED names
into them, leave third-party entries alone, and still build the nested modules through its own registry."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent('''
    import sys, types, warnings

    class MMRegistry:                       # mmcv.utils.Registry's surface, as far as a plugin uses it
        def __init__(self, name):
            self.name, self.module_dict = name, {}
        def register_module(self, name=None, force=False, module=None):
            def _reg(cls):
                key = name or cls.__name__
                if key in self.module_dict and not force:
                    raise KeyError(key + ' is already registered in ' + self.name)
                self.module_dict[key] = cls
                return cls
            return _reg(module) if module is not None else _reg
        def get(self, key):
            return self.module_dict.get(key)

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__path__ = []
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    regs = {n: MMRegistry(n) for n in ('ATTENTION', 'TRANSFORMER_LAYER', 'TRANSFORMER_LAYER_SEQUENCE', 'TRANSFORMER',
                                       'HEADS', 'BBOX_CODERS', 'BBOX_ASSIGNERS', 'MATCH_COST', 'PIPELINES')}
    mod('mmcv'); mod('mmcv.cnn'); mod('mmcv.cnn.bricks')
    mod('mmcv.cnn.bricks.registry', ATTENTION=regs['ATTENTION'], TRANSFORMER_LAYER=regs['TRANSFORMER_LAYER'],
        TRANSFORMER_LAYER_SEQUENCE=regs['TRANSFORMER_LAYER_SEQUENCE'])
    mod('mmdet'); mod('mmdet.models', HEADS=regs['HEADS']); mod('mmdet.models.utils')
    mod('mmdet.models.utils.builder', TRANSFORMER=regs['TRANSFORMER'])
    mod('mmdet.core'); mod('mmdet.core.bbox'); mod('mmdet.core.bbox.match_costs')
    mod('mmdet.core.bbox.builder', BBOX_CODERS=regs['BBOX_CODERS'], BBOX_ASSIGNERS=regs['BBOX_ASSIGNERS'])
    mod('mmdet.core.bbox.match_costs.builder', MATCH_COST=regs['MATCH_COST'])
    mod('mmdet.datasets'); mod('mmdet.datasets.builder', PIPELINES=regs['PIPELINES'])

    # what mmcv / mmdet themselves (and the reference plugin) have registered before this package is imported
    class ThirdPartyMHA: pass
    class ThirdPartyLayer: pass
    class ThirdPartyFocalCost: pass
    class ThirdPartyIoUCost: pass
    class ReferenceHead: pass
    regs['ATTENTION'].register_module(name='MultiheadAttention', module=ThirdPartyMHA)
    regs['TRANSFORMER_LAYER'].register_module(name='DetrTransformerDecoderLayer', module=ThirdPartyLayer)
    regs['MATCH_COST'].register_module(name='FocalLossCost', module=ThirdPartyFocalCost)
    regs['MATCH_COST'].register_module(name='IoUCost', module=ThirdPartyIoUCost)
    regs['HEADS'].register_module(name='Detr3DHead', module=ReferenceHead)

    with warnings.catch_warnings():
        warnings.simplefilter('error')       # a failed mm registration would warn: none may
        import transcar_amd as T
        import transcar_amd.hungarian_assigner_3d, transcar_amd.losses, transcar_amd.radar_pipeline
    from transcar_amd import configs

    # (i) the seven reference-owned names resolve to this package (the reference's own head is replaced)
    owned = [('HEADS', 'Detr3DHead'), ('TRANSFORMER', 'Detr3DTransformer'),
             ('TRANSFORMER_LAYER_SEQUENCE', 'Detr3DTransformerDecoder'), ('ATTENTION', 'Detr3DCrossAtten'),
             ('BBOX_CODERS', 'NMSFreeCoder'), ('BBOX_ASSIGNERS', 'HungarianAssigner3D'), ('MATCH_COST', 'BBox3DL1Cost')]
    for reg, name in owned:
        cls = regs[reg].get(name)
        assert cls is not None and cls.__module__.startswith('transcar_amd.'), (reg, name, cls)
    # (ii) third-party names are untouched
    assert regs['ATTENTION'].get('MultiheadAttention') is ThirdPartyMHA
    assert regs['TRANSFORMER_LAYER'].get('DetrTransformerDecoderLayer') is ThirdPartyLayer
    assert regs['MATCH_COST'].get('FocalLossCost') is ThirdPartyFocalCost
    assert regs['MATCH_COST'].get('IoUCost') is ThirdPartyIoUCost
    exported = {(r, n) for r in regs for n in regs[r].module_dict
                if getattr(regs[r].module_dict[n], '__module__', '').startswith('transcar_amd.')}
    assert exported == set(owned) | {('PIPELINES', 'LoadRadarPointsMultiSweep'), ('PIPELINES', 'BuildRadarFeatures')}, exported
    # (iii) the class mmdet would instantiate builds its nested transformer / coder / assigner / costs through
    # THIS package's registry (the mm registries hold third-party stand-ins that cannot be built)
    head = regs['HEADS'].get('Detr3DHead')(train_cfg=configs.train_cfg_pts,
                                       **{k: v for k, v in configs.head_cfg().items() if k != 'type'})
    layer = head.transformer.decoder.layers[0]
    assert type(layer).__module__ == 'transcar_amd.bricks' and type(layer.attentions[0]).__module__ == 'transcar_amd.bricks'
    assert type(layer.attentions[1]).__name__ == 'Detr3DCrossAtten'
    assert type(head.assigner.cls_cost).__module__ == 'transcar_amd.losses'
    assert len(head.state_dict()) > 300
    print('MM-REGISTRATION-OK')
''')


def test_only_reference_owned_names_reach_the_mm_registries():
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''))
    out = subprocess.run([sys.executable, '-c', SCRIPT], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and 'MM-REGISTRATION-OK' in out.stdout, out.stdout + out.stderr
