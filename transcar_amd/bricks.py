"""Host-side mirrors of the mmcv transformer bricks the reference's config
instantiates (CFG:65-82): same class names, ctor kwargs and state_dict keys
(``attentions.N.attn.in_proj_weight``, ``ffns.0.layers.0.0.weight``,
``norms.N.weight`` ...), so a TransCAR checkpoint loads with strict=True.

The modules own the parameters; their ``forward`` runs the HIP kernels through
``transcar_amd.ops`` (fp32, eval mode).  There is no CPU or eager fallback: a
CPU tensor or a missing library raises ``TransCARHipError``.
"""
import copy

import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from .registry import (ATTENTION, TRANSFORMER_LAYER, build_attention,
                       build_transformer_layer)


class BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = copy.deepcopy(init_cfg)

    def init_weights(self):
        pass


def require_eval(mod):
    if mod.training:
        raise L.TransCARHipError(
            '%s: the HIP path implements eval-mode forward only (dropout '
            'inactive); call .eval() first' % type(mod).__name__)


def qbc_to_bqc(x):
    """[Q,B,C] (reference layout) -> contiguous [B,Q,C]."""
    return x.transpose(0, 1).contiguous()


def mha_view(attn):
    """tc_mha over an nn.MultiheadAttention's parameters."""
    return L.tc_mha(ops.linear_view(attn.in_proj_weight, attn.in_proj_bias),
                    ops.linear_view(attn.out_proj.weight, attn.out_proj.bias))


@ATTENTION.register_module()
class MultiheadAttention(BaseModule):
    """mmcv ``MultiheadAttention`` wrapper."""

    def __init__(self, embed_dims, num_heads, attn_drop=0., proj_drop=0.,
                 dropout_layer=None, init_cfg=None, batch_first=False,
                 **kwargs):
        super().__init__(init_cfg)
        drop_prob = 0.
        if dropout_layer is not None:
            drop_prob = dropout_layer.get('drop_prob', 0.)
        if 'dropout' in kwargs:                # deprecated kwarg in the config
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
        require_eval(self)
        # the decoder's self-attention call pattern: k = q (+pos), v = q
        same = (key is None or key is query) and \
            (value is None or value is query) and \
            (identity is None or identity is query) and \
            (key_pos is None or key_pos is query_pos)
        if not same or attn_mask is not None or key_padding_mask is not None \
                or query_pos is None:
            raise NotImplementedError(
                'MultiheadAttention(HIP): only decoder self-attention '
                '(key=value=identity=query, key_pos=query_pos, no masks) is on '
                'the TransCAR hot path')
        x = qbc_to_bqc(query)
        pos = qbc_to_bqc(query_pos)
        out = ops.self_attn(mha_view(self.attn), x, pos, self.num_heads)
        return out.transpose(0, 1)


class FFN(BaseModule):
    def __init__(self, embed_dims=256, feedforward_channels=1024, num_fcs=2,
                 act_cfg=None, ffn_drop=0., dropout_layer=None,
                 add_identity=True, init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        if num_fcs != 2:
            raise NotImplementedError('FFN(HIP): num_fcs=2 only')
        self.embed_dims = embed_dims
        self.feedforward_channels = feedforward_channels
        self.layers = nn.Sequential(
            nn.Sequential(nn.Linear(embed_dims, feedforward_channels),
                          nn.ReLU(inplace=True), nn.Dropout(ffn_drop)),
            nn.Linear(feedforward_channels, embed_dims),
            nn.Dropout(ffn_drop))
        self.dropout_layer = nn.Identity()
        self.add_identity = add_identity

    def forward(self, x, identity=None):
        require_eval(self)
        x = x.contiguous()
        fc0, fc1 = self.layers[0][0], self.layers[1]
        h = ops.linear(x, fc0.weight, fc0.bias, act=1)
        res = None
        if self.add_identity:
            res = x if identity is None else identity.contiguous()
        return ops.linear(h, fc1.weight, fc1.bias, res=res)


class BaseTransformerLayer(BaseModule):
    def __init__(self, attn_cfgs=None, ffn_cfgs=None, operation_order=None,
                 norm_cfg=None, init_cfg=None, batch_first=False, **kwargs):
        super().__init__(init_cfg)
        ffn_cfgs = dict(ffn_cfgs) if ffn_cfgs else dict(
            embed_dims=256, feedforward_channels=1024, num_fcs=2, ffn_drop=0.)
        for ori, new in dict(feedforward_channels='feedforward_channels',
                             ffn_dropout='ffn_drop',
                             ffn_num_fcs='num_fcs').items():
            if ori in kwargs:
                ffn_cfgs[new] = kwargs[ori]
        self.batch_first = batch_first
        num_attn = operation_order.count('self_attn') + \
            operation_order.count('cross_attn')
        if isinstance(attn_cfgs, dict):
            attn_cfgs = [copy.deepcopy(attn_cfgs) for _ in range(num_attn)]
        assert num_attn == len(attn_cfgs)
        self.num_attn = num_attn
        self.operation_order = tuple(operation_order)
        self.pre_norm = operation_order[0] == 'norm'
        self.attentions = nn.ModuleList()
        index = 0
        for name in operation_order:
            if name in ('self_attn', 'cross_attn'):
                cfg = dict(attn_cfgs[index])
                cfg.setdefault('batch_first', batch_first)
                attn = build_attention(cfg)
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
        self.norms = nn.ModuleList(
            [nn.LayerNorm(self.embed_dims)
             for _ in range(operation_order.count('norm'))])

    def forward(self, query, key=None, value=None, query_pos=None,
                key_pos=None, attn_masks=None, query_key_padding_mask=None,
                key_padding_mask=None, **kwargs):
        require_eval(self)
        if self.pre_norm:
            raise NotImplementedError('pre-norm layers are not on the hot path')
        norm_index = attn_index = ffn_index = 0
        for layer in self.operation_order:
            if layer == 'self_attn':
                query = self.attentions[attn_index](
                    query, query, query, None, query_pos=query_pos,
                    key_pos=query_pos, attn_mask=None,
                    key_padding_mask=query_key_padding_mask, **kwargs)
                attn_index += 1
            elif layer == 'norm':
                n = self.norms[norm_index]
                query = ops.add_layernorm(query.contiguous(), None, n.weight,
                                          n.bias)
                norm_index += 1
            elif layer == 'cross_attn':
                query = self.attentions[attn_index](
                    query, key, value, None, query_pos=query_pos,
                    key_pos=key_pos, attn_mask=None,
                    key_padding_mask=key_padding_mask, **kwargs)
                attn_index += 1
            elif layer == 'ffn':
                query = self.ffns[ffn_index](query, None)
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
        assert set(operation_order) == {'self_attn', 'norm', 'cross_attn', 'ffn'}


class TransformerLayerSequence(BaseModule):
    def __init__(self, transformerlayers=None, num_layers=None, init_cfg=None):
        super().__init__(init_cfg)
        if isinstance(transformerlayers, dict):
            transformerlayers = [copy.deepcopy(transformerlayers)
                                 for _ in range(num_layers)]
        assert len(transformerlayers) == num_layers
        self.num_layers = num_layers
        self.layers = nn.ModuleList(
            [build_transformer_layer(transformerlayers[i])
             for i in range(num_layers)])
        self.embed_dims = self.layers[0].embed_dims
        self.pre_norm = self.layers[0].pre_norm
