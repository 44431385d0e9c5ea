"""oracle/mmcv_bricks.py against stock ``torch.nn``.

The reference's mmdetection3d submodule is empty (/root/reference/.gitmodules:1-3), so the OpenMMLab bricks its config
instantiates (projects/configs/detr3d/detr3d_res101_gridmask.py:65-82) are restated in the oracle with no reference
source to pin them to.  What can be pinned is their published semantics: a post-norm ('self_attn', 'norm',
'cross_attn', 'norm', 'ffn', 'norm') layer of MultiheadAttention / FFN / LayerNorm bricks IS torch's
``nn.TransformerDecoderLayer(norm_first=False, activation='relu')``; the positional-encoding arguments and the
state_dict key names (the ones a TransCAR checkpoint carries) are checked against hand-written formulas.
"""
import math

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import mmcv_bricks as M

E, H, FF = 64, 8, 128


def _layer(order=('self_attn', 'norm', 'cross_attn', 'norm', 'ffn', 'norm'), dropout=0.0):
    return M.DetrTransformerDecoderLayer(
        attn_cfgs=[dict(type='MultiheadAttention', embed_dims=E, num_heads=H, dropout=dropout) for _ in range(2)],
        feedforward_channels=FF, ffn_dropout=dropout, operation_order=order)


def test_registry_shape_and_duplicate_names():
    r = M.Registry('things')

    @r.register_module()
    class A:
        def __init__(self, x=1):
            self.x = x

    assert r.get('A') is A and r.build(dict(type='A', x=3)).x == 3
    with pytest.raises(KeyError):
        r.register_module()(A)
    r.register_module(force=True)(A)
    with pytest.raises(KeyError):
        r.build(dict(type='B'))


def test_state_dict_keys_are_the_checkpoint_keys():
    keys = set(_layer().state_dict())
    for k in ('attentions.0.attn.in_proj_weight', 'attentions.0.attn.in_proj_bias',
              'attentions.0.attn.out_proj.weight', 'attentions.1.attn.out_proj.bias',
              'ffns.0.layers.0.0.weight', 'ffns.0.layers.0.0.bias', 'ffns.0.layers.1.weight',
              'norms.0.weight', 'norms.2.bias'):
        assert k in keys, k
    assert len(keys) == 2 * 4 + 4 + 3 * 2


def test_multihead_attention_is_softmax_attention_plus_identity():
    torch.manual_seed(0)
    m = M.MultiheadAttention(E, H, dropout=0.0).eval()
    q, pos = torch.randn(10, 2, E), torch.randn(10, 2, E)
    k, kpos = torch.randn(7, 2, E), torch.randn(7, 2, E)
    with torch.no_grad():
        out = m(q, k, k, query_pos=pos, key_pos=kpos)
    w, b = m.attn.in_proj_weight, m.attn.in_proj_bias
    qq = F.linear(q + pos, w[:E], b[:E])
    kk = F.linear(k + kpos, w[E:2 * E], b[E:2 * E])
    vv = F.linear(k, w[2 * E:], b[2 * E:])           # value carries no positional term

    def heads(x):
        return x.reshape(x.shape[0], 2 * H, E // H).transpose(0, 1)
    a = torch.softmax(heads(qq) @ heads(kk).transpose(1, 2) / math.sqrt(E // H), -1) @ heads(vv)
    ref = q + F.linear(a.transpose(0, 1).reshape(10, 2, E), m.attn.out_proj.weight, m.attn.out_proj.bias)
    assert float((out - ref).abs().max()) < 1e-5
    # key defaults to the query and the query's positional term is reused when shapes agree (self-attention)
    out = m(q, query_pos=pos)
    ref = m(q, q, q, query_pos=pos, key_pos=pos)
    assert torch.equal(out, ref)
    # a cross-attention key of another length gets no positional term
    assert torch.equal(m(q, k, k, query_pos=pos), m(q, k, k, query_pos=pos, key_pos=None))


def test_ffn_is_linear_relu_linear_plus_identity():
    torch.manual_seed(1)
    f = M.FFN(E, FF).eval()
    x, idt = torch.randn(5, 3, E), torch.randn(5, 3, E)
    l0, l1 = f.layers[0][0], f.layers[1]
    ref = F.linear(F.relu(F.linear(x, l0.weight, l0.bias)), l1.weight, l1.bias)
    assert float((f(x) - (x + ref)).abs().max()) < 1e-6
    assert float((f(x, idt) - (idt + ref)).abs().max()) < 1e-6
    assert float((M.FFN(E, FF, add_identity=False).eval()(x) - x).abs().max()) > 1e-3


def test_post_norm_layer_is_torch_transformer_decoder_layer():
    torch.manual_seed(2)
    lay = _layer().eval()
    ref = nn.TransformerDecoderLayer(E, H, dim_feedforward=FF, dropout=0.0, activation='relu', norm_first=False).eval()
    sd = lay.state_dict()
    ref.load_state_dict({
        'self_attn.in_proj_weight': sd['attentions.0.attn.in_proj_weight'],
        'self_attn.in_proj_bias': sd['attentions.0.attn.in_proj_bias'],
        'self_attn.out_proj.weight': sd['attentions.0.attn.out_proj.weight'],
        'self_attn.out_proj.bias': sd['attentions.0.attn.out_proj.bias'],
        'multihead_attn.in_proj_weight': sd['attentions.1.attn.in_proj_weight'],
        'multihead_attn.in_proj_bias': sd['attentions.1.attn.in_proj_bias'],
        'multihead_attn.out_proj.weight': sd['attentions.1.attn.out_proj.weight'],
        'multihead_attn.out_proj.bias': sd['attentions.1.attn.out_proj.bias'],
        'linear1.weight': sd['ffns.0.layers.0.0.weight'], 'linear1.bias': sd['ffns.0.layers.0.0.bias'],
        'linear2.weight': sd['ffns.0.layers.1.weight'], 'linear2.bias': sd['ffns.0.layers.1.bias'],
        'norm1.weight': sd['norms.0.weight'], 'norm1.bias': sd['norms.0.bias'],
        'norm2.weight': sd['norms.1.weight'], 'norm2.bias': sd['norms.1.bias'],
        'norm3.weight': sd['norms.2.weight'], 'norm3.bias': sd['norms.2.bias']})
    q, mem = torch.randn(12, 2, E), torch.randn(9, 2, E)
    with torch.no_grad():
        a = lay(q, mem, mem)
        b = ref(q, mem)
    assert float((a - b).abs().max()) < 2e-5


def test_layer_adds_query_pos_to_queries_and_self_attention_keys_only():
    torch.manual_seed(3)
    lay = _layer().eval()
    q, pos, mem = torch.randn(6, 1, E), torch.randn(6, 1, E), torch.randn(4, 1, E)
    with torch.no_grad():
        out = lay(q, mem, mem, query_pos=pos)
        sa, ca = lay.attentions
        x = lay.norms[0](sa(q, q, q, query_pos=pos, key_pos=pos))
        x = lay.norms[1](ca(x, mem, mem, query_pos=pos, key_pos=None))
        x = lay.norms[2](lay.ffns[0](x))
    assert float((out - x).abs().max()) < 1e-6


def test_layer_sequence_builds_independent_copies():
    seq = M.TransformerLayerSequence(
        transformerlayers=dict(type='DetrTransformerDecoderLayer',
                               attn_cfgs=[dict(type='MultiheadAttention', embed_dims=E, num_heads=H, dropout=0.1)] * 2,
                               feedforward_channels=FF, ffn_dropout=0.1,
                               operation_order=('self_attn', 'norm', 'cross_attn', 'norm', 'ffn', 'norm')),
        num_layers=3)
    assert seq.num_layers == 3 and len(seq.layers) == 3 and seq.embed_dims == E and not seq.pre_norm
    assert seq.layers[0].attentions[0].attn.in_proj_weight.data_ptr() != \
        seq.layers[1].attentions[0].attn.in_proj_weight.data_ptr()
    # the config's deprecated `dropout` kwarg sets both the attention dropout and the residual dropout (mmcv)
    a = seq.layers[0].attentions[0]
    assert a.attn.dropout == 0.1 and isinstance(a.dropout_layer, nn.Dropout) and a.dropout_layer.p == 0.1
    assert seq.layers[0].ffns[0].layers[0][2].p == 0.1
