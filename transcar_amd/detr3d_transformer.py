"""Host-side mirror of the reference's
projects/mmdet3d_plugin/models/utils/detr3d_transformer.py: same registry
names, ctor kwargs, forward signatures and parameter names; the arithmetic
runs in the HIP library (fp32, eval mode).

  Detr3DTransformer         XFMR:35-139
  Detr3DTransformerDecoder  XFMR:142-214
  Detr3DCrossAtten          XFMR:217-378  (+ feature_sampling XFMR:381-422)
"""
import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from .bricks import (BaseModule, TransformerLayerSequence, qbc_to_bqc,
                     require_eval)
from .registry import (ATTENTION, TRANSFORMER, TRANSFORMER_LAYER_SEQUENCE,
                       build_transformer_layer_sequence)


def pos_encoder_view(seq):
    """tc_pos_encoder over nn.Sequential(Linear, LN, ReLU, Linear, LN, ReLU)."""
    return L.tc_pos_encoder(ops.linear_view(seq[0].weight, seq[0].bias),
                            ops.lnorm_view(seq[1].weight, seq[1].bias),
                            ops.linear_view(seq[3].weight, seq[3].bias),
                            ops.lnorm_view(seq[4].weight, seq[4].bias))


def reg_branch_forward(branch, x):
    """Linear-ReLU-Linear-ReLU-Linear (HEAD:208-213) through the HIP GEMM."""
    h = ops.linear(x, branch[0].weight, branch[0].bias, act=1)
    h = ops.linear(h, branch[2].weight, branch[2].bias, act=1)
    return ops.linear(h, branch[4].weight, branch[4].bias)


class FeatureCache:
    """NCHW -> NHWC conversion of the FPN maps, done once per forward and
    shared by the 6 decoder layers (``value`` is the same list object)."""

    def __init__(self):
        self._key = None
        self._nhwc = None

    def get(self, mlvl_feats):
        key = tuple((f.data_ptr(), f._version, tuple(f.shape))
                    for f in mlvl_feats)
        if key != self._key:
            self._nhwc = [ops.to_nhwc(f) for f in mlvl_feats]
            self._key = key
        return self._nhwc


_FEATS = FeatureCache()


@ATTENTION.register_module(export=True)
class Detr3DCrossAtten(BaseModule):
    """Camera cross-attention of DETR3D (XFMR:217-378)."""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=5,
                 num_cams=6, im2col_step=64, pc_range=None, dropout=0.1,
                 norm_cfg=None, init_cfg=None, batch_first=False):
        super().__init__(init_cfg)
        if embed_dims % num_heads != 0:
            raise ValueError('embed_dims must be divisible by num_heads, '
                             'but got %d and %d' % (embed_dims, num_heads))
        if num_points != 1:
            raise NotImplementedError(
                'Detr3DCrossAtten(HIP): num_points=1 (the TransCAR configs, '
                'CFG:75)')
        self.norm_cfg = norm_cfg
        self.dropout = nn.Dropout(dropout)
        self.pc_range = pc_range
        self.im2col_step = im2col_step
        self.embed_dims = embed_dims
        self.num_levels = num_levels
        self.num_heads = num_heads
        self.num_points = num_points
        self.num_cams = num_cams
        self.attention_weights = nn.Linear(embed_dims,
                                           num_cams * num_levels * num_points)
        self.output_proj = nn.Linear(embed_dims, embed_dims)
        self.position_encoder = nn.Sequential(
            nn.Linear(3, embed_dims), nn.LayerNorm(embed_dims),
            nn.ReLU(inplace=True),
            nn.Linear(embed_dims, embed_dims), nn.LayerNorm(embed_dims),
            nn.ReLU(inplace=True))
        self.batch_first = batch_first
        self.init_weight()

    def init_weight(self):
        """XFMR:297-300."""
        nn.init.constant_(self.attention_weights.weight, 0.)
        nn.init.constant_(self.attention_weights.bias, 0.)
        nn.init.xavier_uniform_(self.output_proj.weight)
        nn.init.constant_(self.output_proj.bias, 0.)

    def forward(self, query, key, value, residual=None, query_pos=None,
                key_padding_mask=None, reference_points=None,
                spatial_shapes=None, level_start_index=None, **kwargs):
        """query [Q,B,C]; value = list of [B,N,C,H,W]; reference_points
        [B,Q,3] normalised; kwargs['img_metas'].  Returns [Q,B,C]."""
        require_eval(self)
        if residual is not None:
            raise NotImplementedError('residual must be None (XFMR:351-352)')
        img_metas = kwargs['img_metas']
        feats = _FEATS.get(value)
        q = qbc_to_bqc(query)
        pos = qbc_to_bqc(query_pos) if query_pos is not None \
            else torch.zeros_like(q)
        l2i = ops.lidar2img_tensor(img_metas, q.device)
        img_hw = img_metas[0]['img_shape'][0][:2]
        out = ops.cross_atten(
            ops.linear_view(self.attention_weights.weight,
                            self.attention_weights.bias),
            ops.linear_view(self.output_proj.weight, self.output_proj.bias),
            pos_encoder_view(self.position_encoder), feats, q, pos, l2i,
            reference_points.contiguous(), self.pc_range, img_hw,
            self.num_cams)
        return out.transpose(0, 1)


@TRANSFORMER_LAYER_SEQUENCE.register_module(export=True)
class Detr3DTransformerDecoder(TransformerLayerSequence):
    """XFMR:142-214."""

    def __init__(self, *args, return_intermediate=False, **kwargs):
        super().__init__(*args, **kwargs)
        self.return_intermediate = return_intermediate

    def forward(self, query, *args, reference_points=None, reg_branches=None,
                **kwargs):
        output = query
        intermediate, intermediate_reference_points = [], []
        for lid, layer in enumerate(self.layers):
            output = layer(output, *args, reference_points=reference_points,
                           **kwargs)
            output = output.permute(1, 0, 2)
            if reg_branches is not None:
                tmp = reg_branch_forward(reg_branches[lid],
                                         output.contiguous())
                assert reference_points.shape[-1] == 3
                reference_points = ops.refine_reference(
                    tmp, reference_points.contiguous())
            output = output.permute(1, 0, 2)
            if self.return_intermediate:
                intermediate.append(output)
                intermediate_reference_points.append(reference_points)
        if self.return_intermediate:
            return torch.stack(intermediate), torch.stack(
                intermediate_reference_points)
        return output, reference_points


@TRANSFORMER.register_module(export=True)
class Detr3DTransformer(BaseModule):
    """XFMR:35-139."""

    def __init__(self, num_feature_levels=4, num_cams=6,
                 two_stage_num_proposals=300, decoder=None, **kwargs):
        super().__init__(**kwargs)
        self.decoder = build_transformer_layer_sequence(decoder)
        self.embed_dims = self.decoder.embed_dims
        self.num_feature_levels = num_feature_levels
        self.num_cams = num_cams
        self.two_stage_num_proposals = two_stage_num_proposals
        self.init_layers()

    def init_layers(self):
        self.reference_points = nn.Linear(self.embed_dims, 3)

    def init_weights(self):
        """XFMR:65-73."""
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, Detr3DCrossAtten):
                m.init_weight()
        nn.init.xavier_uniform_(self.reference_points.weight)
        nn.init.constant_(self.reference_points.bias, 0.)

    def forward(self, mlvl_feats, query_embed, reg_branches=None, **kwargs):
        """Returns inter_states [L,Q,B,C], init_reference [B,Q,3],
        inter_references [L,B,Q,3] (XFMR:117-139)."""
        require_eval(self)
        assert query_embed is not None
        bs = mlvl_feats[0].size(0)
        query_pos, query = torch.split(query_embed, self.embed_dims, dim=1)
        query_pos = query_pos.unsqueeze(0).expand(bs, -1, -1).contiguous()
        query = query.unsqueeze(0).expand(bs, -1, -1).contiguous()
        reference_points = ops.linear(query_pos, self.reference_points.weight,
                                      self.reference_points.bias, act=2)
        init_reference_out = reference_points
        inter_states, inter_references = self.decoder(
            query=query.permute(1, 0, 2), key=None, value=mlvl_feats,
            query_pos=query_pos.permute(1, 0, 2),
            reference_points=reference_points, reg_branches=reg_branches,
            **kwargs)
        return inter_states, init_reference_out, inter_references
