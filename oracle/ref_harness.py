"""Import the reference's OWN hot-path files, unmodified, in the authoring
container (TEST INFRASTRUCTURE).

The reference needs mmcv / mmdet / mmdet3d / nuscenes-devkit / pyquaternion,
none of which is installed or vendored.  This module injects stub
``sys.modules`` entries that provide exactly the symbols the four reference
files read, then file-loads

  projects/mmdet3d_plugin/core/bbox/util.py                  (UTIL)
  projects/mmdet3d_plugin/core/bbox/coders/nms_free_coder.py (CODER)
  projects/mmdet3d_plugin/models/utils/detr3d_transformer.py (XFMR)
  projects/mmdet3d_plugin/models/dense_heads/detr3d_head.py  (HEAD)

from /root/reference so that all reference-owned arithmetic runs as written.
It is used only by tests/golden/make_golden.py and by the tests that are
skipped when /root/reference is absent (it is absent on the GPU box).
No reference source is copied: the files are executed where they lie.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

from . import mmcv_bricks as B
from . import mmdet_bricks as MD

REF_ROOT = os.environ.get('TRANSCAR_REFERENCE', '/root/reference')
PLUGIN = os.path.join(REF_ROOT, 'projects', 'mmdet3d_plugin')


def available():
    return os.path.isfile(os.path.join(
        PLUGIN, 'models', 'dense_heads', 'detr3d_head.py'))


# --------------------------------------------------------------------------
# fake nuScenes devkit: the head pulls radar sweeps + calibration through it
# (HEAD:27, 301-375).  The fake serves whatever RADAR_FRAME holds.
# --------------------------------------------------------------------------
RADAR_CHANNELS = ['RADAR_FRONT', 'RADAR_FRONT_LEFT', 'RADAR_FRONT_RIGHT',
                  'RADAR_BACK_LEFT', 'RADAR_BACK_RIGHT']

#: set by the caller before head.forward: dict(points={chan: [18,n] f64},
#: times={chan: [1,n]}, radar_rot={chan: wxyz}, lidar_rot=wxyz)
RADAR_FRAME = {}


class _FakeNuScenes:
    def __init__(self, version=None, dataroot=None, verbose=False):
        pass

    def get(self, table, token):
        if table == 'sample':
            data = {c: 'sd:' + c for c in RADAR_CHANNELS}
            data['LIDAR_TOP'] = 'sd:LIDAR_TOP'
            return {'token': token, 'data': data}
        if table == 'sample_data':
            return {'calibrated_sensor_token': 'cs:' + token.split(':')[1]}
        if table == 'calibrated_sensor':
            chan = token.split(':')[1]
            if chan == 'LIDAR_TOP':
                return {'rotation': list(RADAR_FRAME['lidar_rot'])}
            return {'rotation': list(RADAR_FRAME['radar_rot'][chan])}
        raise KeyError(table)


class _FakeRadarPointCloud:
    def __init__(self, points):
        self.points = points

    @classmethod
    def from_file_multisweep(cls, nusc, sample_rec, chan, ref_chan,
                             nsweeps=5, min_distance=1.0):
        return (cls(np.array(RADAR_FRAME['points'][chan], dtype=np.float64)),
                np.array(RADAR_FRAME['times'][chan], dtype=np.float64))


class _Quaternion:
    """pyquaternion.Quaternion(wxyz).rotation_matrix (unit-normalised)."""

    def __init__(self, q):
        self.q = np.asarray(q, dtype=np.float64)

    @property
    def rotation_matrix(self):
        w, x, y, z = self.q / np.linalg.norm(self.q)
        return np.array([
            [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
            [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
            [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


# --------------------------------------------------------------------------
# mmdet pieces
# --------------------------------------------------------------------------
def inverse_sigmoid(x, eps=1e-5):
    x = x.clamp(min=0, max=1)
    x1 = x.clamp(min=eps)
    x2 = (1 - x).clamp(min=eps)
    return torch.log(x1 / x2)


def multi_apply(func, *args, **kwargs):
    from functools import partial
    pfunc = partial(func, **kwargs) if kwargs else func
    return tuple(map(list, zip(*map(pfunc, *args))))


def reduce_mean(tensor):
    return tensor


class _LossStub(nn.Module):
    def __init__(self, use_sigmoid=False, **kw):
        super().__init__()
        self.use_sigmoid = use_sigmoid
        self.kw = kw


class DETRHead(B.BaseModule):
    """Sets exactly the attributes ``Detr3DHead`` reads."""

    def __init__(self, num_classes, in_channels, num_query=100,
                 num_reg_fcs=2, transformer=None, sync_cls_avg_factor=False,
                 positional_encoding=None, loss_cls=None, loss_bbox=None,
                 loss_iou=None, train_cfg=None, test_cfg=None, init_cfg=None,
                 **kwargs):
        super().__init__(init_cfg)
        self.bg_cls_weight = 0
        self.sync_cls_avg_factor = sync_cls_avg_factor
        self.num_query = num_query
        self.num_classes = num_classes
        self.in_channels = in_channels
        self.num_reg_fcs = num_reg_fcs
        self.train_cfg = train_cfg
        self.test_cfg = test_cfg
        self.fp16_enabled = False
        self.loss_cls = MD.build_loss(loss_cls)
        self.loss_bbox = MD.build_loss(loss_bbox)
        self.loss_iou = MD.build_loss(loss_iou)
        if train_cfg:
            self.assigner = B.BBOX_ASSIGNERS.build(train_cfg['assigner'])
            self.sampler = MD.PseudoSampler()
        self.cls_out_channels = num_classes if self.loss_cls.use_sigmoid \
            else num_classes + 1
        self.transformer = B.TRANSFORMER.build(transformer)
        self.embed_dims = self.transformer.embed_dims
        self._init_layers()


class BaseBBoxCoder:
    def __init__(self, **kwargs):
        pass


def build_bbox_coder(cfg):
    return B.BBOX_CODERS.build(cfg)


def _identity_decorator(*a, **k):
    def deco(f):
        return f
    return deco


_LOADED = {}


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _load(dotted, path):
    spec = importlib.util.spec_from_file_location(dotted, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[dotted] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    """Returns a namespace with the reference's UTIL, CODER, XFMR, HEAD."""
    if _LOADED:
        return types.SimpleNamespace(**_LOADED)
    if not available():
        raise RuntimeError('reference tree not present at %s' % REF_ROOT)
    sys.dont_write_bytecode = True       # reference mount is read-only
    # .cuda() is hard-coded at HEAD:523,526,540; there is no GPU here
    torch.Tensor.cuda = lambda self, *a, **k: self

    _mod('mmcv')
    _mod('mmcv.cnn', Linear=nn.Linear, xavier_init=B.xavier_init,
         constant_init=B.constant_init,
         bias_init_with_prob=B.bias_init_with_prob)
    _mod('mmcv.cnn.bricks')
    _mod('mmcv.cnn.bricks.registry', ATTENTION=B.ATTENTION,
         TRANSFORMER_LAYER_SEQUENCE=B.TRANSFORMER_LAYER_SEQUENCE)
    _mod('mmcv.cnn.bricks.transformer',
         MultiScaleDeformableAttention=B.MultiScaleDeformableAttention,
         TransformerLayerSequence=B.TransformerLayerSequence,
         build_transformer_layer_sequence=B.build_transformer_layer_sequence)
    _mod('mmcv.runner', force_fp32=_identity_decorator,
         auto_fp16=_identity_decorator, BaseModule=B.BaseModule)
    _mod('mmcv.runner.base_module', BaseModule=B.BaseModule)
    _mod('mmdet')
    _mod('mmdet.core', multi_apply=multi_apply, reduce_mean=reduce_mean)
    _mod('mmdet.core.bbox', BaseBBoxCoder=BaseBBoxCoder)
    _mod('mmdet.core.bbox.assigners', AssignResult=MD.AssignResult, BaseAssigner=MD.BaseAssigner)
    _mod('mmdet.core.bbox.match_costs', build_match_cost=MD.build_match_cost)
    _mod('mmdet.core.bbox.match_costs.builder', MATCH_COST=B.MATCH_COST)
    _mod('mmdet.core.bbox.builder', BBOX_CODERS=B.BBOX_CODERS,
         BBOX_ASSIGNERS=B.BBOX_ASSIGNERS)
    _mod('mmdet.models', HEADS=B.HEADS)
    _mod('mmdet.models.utils')
    _mod('mmdet.models.utils.builder', TRANSFORMER=B.TRANSFORMER)
    _mod('mmdet.models.utils.transformer', inverse_sigmoid=inverse_sigmoid)
    _mod('mmdet.models.dense_heads', DETRHead=DETRHead)
    _mod('mmdet3d')
    _mod('mmdet3d.core')
    _mod('mmdet3d.core.bbox')
    _mod('mmdet3d.core.bbox.coders', build_bbox_coder=build_bbox_coder)
    _mod('nuscenes')
    _mod('nuscenes.nuscenes', NuScenes=_FakeNuScenes)
    _mod('nuscenes.utils')
    _mod('nuscenes.utils.data_classes', RadarPointCloud=_FakeRadarPointCloud)
    _mod('pyquaternion', Quaternion=_Quaternion)
    for pkg in ['projects', 'projects.mmdet3d_plugin',
                'projects.mmdet3d_plugin.core',
                'projects.mmdet3d_plugin.core.bbox',
                'projects.mmdet3d_plugin.core.bbox.coders',
                'projects.mmdet3d_plugin.core.bbox.assigners',
                'projects.mmdet3d_plugin.core.bbox.match_costs',
                'projects.mmdet3d_plugin.models',
                'projects.mmdet3d_plugin.models.utils',
                'projects.mmdet3d_plugin.models.dense_heads']:
        _mod(pkg)

    P = 'projects.mmdet3d_plugin.'
    _LOADED['UTIL'] = _load(P + 'core.bbox.util',
                            os.path.join(PLUGIN, 'core/bbox/util.py'))
    _LOADED['CODER'] = _load(
        P + 'core.bbox.coders.nms_free_coder',
        os.path.join(PLUGIN, 'core/bbox/coders/nms_free_coder.py'))
    _LOADED['COST'] = _load(
        P + 'core.bbox.match_costs.match_cost',
        os.path.join(PLUGIN, 'core/bbox/match_costs/match_cost.py'))
    _LOADED['ASSIGN'] = _load(
        P + 'core.bbox.assigners.hungarian_assigner_3d',
        os.path.join(PLUGIN, 'core/bbox/assigners/hungarian_assigner_3d.py'))
    _LOADED['XFMR'] = _load(
        P + 'models.utils.detr3d_transformer',
        os.path.join(PLUGIN, 'models/utils/detr3d_transformer.py'))
    _LOADED['HEAD'] = _load(
        P + 'models.dense_heads.detr3d_head',
        os.path.join(PLUGIN, 'models/dense_heads/detr3d_head.py'))
    return types.SimpleNamespace(**_LOADED)


class GtBoxes:
    """Stand-in for mmdet3d LiDARInstance3DBoxes as the loss reads it
    (HEAD:963-965): ``tensor`` [n,9] bottom-centre boxes, ``gravity_center``."""

    def __init__(self, tensor):
        self.tensor = tensor

    @property
    def gravity_center(self):
        c = self.tensor[:, :3].clone()
        c[:, 2] = c[:, 2] + self.tensor[:, 5] * 0.5
        return c


def build_reference_head(head_cfg, train_cfg=None):
    """Instantiate the reference's Detr3DHead from a ``pts_bbox_head`` dict."""
    ref = load_reference()
    cfg = dict(head_cfg)
    cfg.pop('type', None)
    if train_cfg is not None:
        cfg['train_cfg'] = train_cfg
    head = ref.HEAD.Detr3DHead(**cfg)
    head.eval()
    return head
