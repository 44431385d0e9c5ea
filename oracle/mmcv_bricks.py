"""Restatement of the un-vendored OpenMMLab bricks the reference's hot path
instantiates (TEST INFRASTRUCTURE, used only by ``ref_harness``).

The reference's ``mmdetection3d`` submodule is empty and un-pinned
(/root/reference/.gitmodules:1-3), so mmcv / mmdet source is not available.
These classes restate the published semantics of the bricks configured at
projects/configs/detr3d/detr3d_res101_gridmask.py:65-82 (SURVEY.md Appendix
B): parity for this file is UNPINNED by the reference; it is cross-checked
against stock ``torch.nn`` modules (``nn.TransformerDecoderLayer`` for the
post-norm layer) in tests/test_oracle_bricks.py.

Sub-module names (``attentions``, ``attn``, ``ffns``, ``layers``, ``norms``)
follow mmcv so that state_dict keys equal those of a TransCAR checkpoint.
"""
import copy
import math

import torch
import torch.nn as nn


class Registry:
    """name -> class map with mmcv's ``register_module()`` decorator shape."""

    def __init__(self, name):
        self.name = name
        self.module_dict = {}

    def register_module(self, name=None, force=False, module=None):
        def _reg(cls):
            key = name or cls.__name__
            if not force and key in self.module_dict:       # mmcv: a second class under one name is an error
                raise KeyError('%s is already registered in %s' % (key, self.name))
            self.module_dict[key] = cls
            return cls
        if module is not None:
            return _reg(module)
        return _reg

    def get(self, key):
        return self.module_dict[key]

    def build(self, cfg, **default_args):
        cfg = dict(cfg)
        for k, v in default_args.items():
            cfg.setdefault(k, v)
        typ = cfg.pop('type')
        cls = self.get(typ) if isinstance(typ, str) else typ
        return cls(**cfg)


ATTENTION = Registry('attention')
TRANSFORMER_LAYER = Registry('transformerLayer')
TRANSFORMER_LAYER_SEQUENCE = Registry('transformer-layers sequence')
TRANSFORMER = Registry('Transformer')
HEADS = Registry('heads')
BBOX_CODERS = Registry('bbox_coder')
BBOX_ASSIGNERS = Registry('bbox_assigner')
MATCH_COST = Registry('Match Cost')
LOSSES = Registry('loss')
DETECTORS = Registry('detectors')


class BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = copy.deepcopy(init_cfg)

    def init_weights(self):
        pass


def xavier_init(module, gain=1, bias=0, distribution='normal'):
    if hasattr(module, 'weight') and module.weight is not None:
        if distribution == 'uniform':
            nn.init.xavier_uniform_(module.weight, gain=gain)
        else:
            nn.init.xavier_normal_(module.weight, gain=gain)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def constant_init(module, val, bias=0):
    if hasattr(module, 'weight') and module.weight is not None:
        nn.init.constant_(module.weight, val)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def bias_init_with_prob(prior_prob):
    return float(-math.log((1 - prior_prob) / prior_prob))


@ATTENTION.register_module()
class MultiheadAttention(BaseModule):
    """mmcv wrapper around nn.MultiheadAttention."""

    def __init__(self, embed_dims, num_heads, attn_drop=0., proj_drop=0.,
                 dropout_layer=None, init_cfg=None, batch_first=False,
                 **kwargs):
        super().__init__(init_cfg)
        drop_prob = 0.
        if dropout_layer is not None:
            drop_prob = dropout_layer.get('drop_prob', 0.)
        if 'dropout' in kwargs:          # deprecated kwarg used by the config
            attn_drop = kwargs['dropout']
            drop_prob = kwargs.pop('dropout')
        self.embed_dims = embed_dims
        self.num_heads = num_heads
        self.batch_first = batch_first
        self.attn = nn.MultiheadAttention(embed_dims, num_heads, attn_drop,
                                          **kwargs)
        self.proj_drop = nn.Dropout(proj_drop)
        self.dropout_layer = nn.Dropout(drop_prob) if drop_prob > 0 \
            else nn.Identity()

    def forward(self, query, key=None, value=None, identity=None,
                query_pos=None, key_pos=None, attn_mask=None,
                key_padding_mask=None, **kwargs):
        if key is None:
            key = query
        if value is None:
            value = key
        if identity is None:
            identity = query
        if key_pos is None:
            if query_pos is not None and query_pos.shape == key.shape:
                key_pos = query_pos
        if query_pos is not None:
            query = query + query_pos
        if key_pos is not None:
            key = key + key_pos
        out = self.attn(query=query, key=key, value=value,
                        attn_mask=attn_mask,
                        key_padding_mask=key_padding_mask)[0]
        return identity + self.dropout_layer(self.proj_drop(out))


class FFN(BaseModule):
    def __init__(self, embed_dims=256, feedforward_channels=1024, num_fcs=2,
                 act_cfg=None, ffn_drop=0., dropout_layer=None,
                 add_identity=True, init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        assert num_fcs >= 2
        self.embed_dims = embed_dims
        layers = []
        in_channels = embed_dims
        for _ in range(num_fcs - 1):
            layers.append(nn.Sequential(
                nn.Linear(in_channels, feedforward_channels),
                nn.ReLU(inplace=True), nn.Dropout(ffn_drop)))
            in_channels = feedforward_channels
        layers.append(nn.Linear(feedforward_channels, embed_dims))
        layers.append(nn.Dropout(ffn_drop))
        self.layers = nn.Sequential(*layers)
        self.dropout_layer = nn.Identity()
        self.add_identity = add_identity

    def forward(self, x, identity=None):
        out = self.layers(x)
        if not self.add_identity:
            return self.dropout_layer(out)
        if identity is None:
            identity = x
        return identity + self.dropout_layer(out)


class BaseTransformerLayer(BaseModule):
    """Post-/pre-norm layer walking ``operation_order`` (mmcv semantics)."""

    def __init__(self, attn_cfgs=None, ffn_cfgs=None, operation_order=None,
                 norm_cfg=None, init_cfg=None, batch_first=False, **kwargs):
        super().__init__(init_cfg)
        ffn_cfgs = dict(ffn_cfgs) if ffn_cfgs else dict(
            embed_dims=256, feedforward_channels=1024, num_fcs=2, ffn_drop=0.)
        deprecated = dict(feedforward_channels='feedforward_channels',
                          ffn_dropout='ffn_drop', ffn_num_fcs='num_fcs')
        for ori, new in deprecated.items():
            if ori in kwargs:
                ffn_cfgs[new] = kwargs[ori]
        self.batch_first = batch_first
        num_attn = operation_order.count('self_attn') + \
            operation_order.count('cross_attn')
        if isinstance(attn_cfgs, dict):
            attn_cfgs = [copy.deepcopy(attn_cfgs) for _ in range(num_attn)]
        assert num_attn == len(attn_cfgs)
        self.num_attn = num_attn
        self.operation_order = operation_order
        self.pre_norm = operation_order[0] == 'norm'
        self.attentions = nn.ModuleList()
        index = 0
        for name in operation_order:
            if name in ('self_attn', 'cross_attn'):
                cfg = dict(attn_cfgs[index])
                cfg.setdefault('batch_first', batch_first)
                attn = ATTENTION.build(cfg)
                attn.operation_name = name
                self.attentions.append(attn)
                index += 1
        self.embed_dims = self.attentions[0].embed_dims
        self.ffns = nn.ModuleList()
        for _ in range(operation_order.count('ffn')):
            cfg = dict(ffn_cfgs)
            cfg.pop('type', None)
            cfg['embed_dims'] = self.embed_dims
            self.ffns.append(FFN(**cfg))
        self.norms = nn.ModuleList()
        for _ in range(operation_order.count('norm')):
            self.norms.append(nn.LayerNorm(self.embed_dims))

    def forward(self, query, key=None, value=None, query_pos=None,
                key_pos=None, attn_masks=None, query_key_padding_mask=None,
                key_padding_mask=None, **kwargs):
        norm_index = attn_index = ffn_index = 0
        identity = query
        if attn_masks is None:
            attn_masks = [None for _ in range(self.num_attn)]
        for layer in self.operation_order:
            if layer == 'self_attn':
                temp_key = temp_value = query
                query = self.attentions[attn_index](
                    query, temp_key, temp_value,
                    identity if self.pre_norm else None,
                    query_pos=query_pos, key_pos=query_pos,
                    attn_mask=attn_masks[attn_index],
                    key_padding_mask=query_key_padding_mask, **kwargs)
                attn_index += 1
                identity = query
            elif layer == 'norm':
                query = self.norms[norm_index](query)
                norm_index += 1
            elif layer == 'cross_attn':
                query = self.attentions[attn_index](
                    query, key, value,
                    identity if self.pre_norm else None,
                    query_pos=query_pos, key_pos=key_pos,
                    attn_mask=attn_masks[attn_index],
                    key_padding_mask=key_padding_mask, **kwargs)
                attn_index += 1
                identity = query
            elif layer == 'ffn':
                query = self.ffns[ffn_index](
                    query, identity if self.pre_norm else None)
                ffn_index += 1
        return query


@TRANSFORMER_LAYER.register_module()
class DetrTransformerDecoderLayer(BaseTransformerLayer):
    def __init__(self, attn_cfgs, feedforward_channels, ffn_dropout=0.0,
                 operation_order=None, act_cfg=None, norm_cfg=None,
                 ffn_num_fcs=2, **kwargs):
        super().__init__(attn_cfgs=attn_cfgs,
                         feedforward_channels=feedforward_channels,
                         ffn_dropout=ffn_dropout,
                         operation_order=operation_order,
                         ffn_num_fcs=ffn_num_fcs, **kwargs)
        assert len(operation_order) == 6


class TransformerLayerSequence(BaseModule):
    def __init__(self, transformerlayers=None, num_layers=None,
                 init_cfg=None):
        super().__init__(init_cfg)
        if isinstance(transformerlayers, dict):
            transformerlayers = [copy.deepcopy(transformerlayers)
                                 for _ in range(num_layers)]
        self.num_layers = num_layers
        self.layers = nn.ModuleList()
        for i in range(num_layers):
            self.layers.append(TRANSFORMER_LAYER.build(transformerlayers[i]))
        self.embed_dims = self.layers[0].embed_dims
        self.pre_norm = self.layers[0].pre_norm


def build_transformer_layer_sequence(cfg):
    return TRANSFORMER_LAYER_SEQUENCE.build(cfg)


class MultiScaleDeformableAttention(BaseModule):
    """Only imported for an isinstance() check (XFMR:9,71)."""
